#!/usr/bin/env python3
"""MI355X counterpart of the reference's unconditional pixel-space trainer.

Same command line and YAML schema as /root/reference/train_uncond_dpm.py (``--cfg <yaml>`` with sections
model{..., unet{...}}, data, trainer, sampler); the recipe it reproduces (reference lines in brackets):
global batch split across ranks [:138-143 split_batches], gradient accumulation [:262-280], AdamW(lr,
wd=1e-4) [:178-179], warm-up / decay lambda [:169-177], clip-norm 1.0 [:292], EMA(beta .9996,
update_after_step, update_every, power 2/3) on rank 0 [:187-189, 308-310], checkpoint dict layout
{'step','model','opt','lr_scheduler','ema','scaler'} in ``results_folder/model-{milestone}.pt``
[:207-239], periodic sample grid [:315-333].  Launch: ``python train_uncond_dpm.py --cfg X`` (1 GPU) or
``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train_uncond_dpm.py --cfg X``.

What differs: one process per GPU with torch.distributed/RCCL instead of HF accelerate; the
optimiser state lives in flat buffers driven by the fused HIP kernel (so 'opt' in the checkpoint holds
{'exp_avg','exp_avg_sq','step'} flat tensors); gradients ARE averaged across ranks (the reference
bypasses DDP.forward and never synchronises them); images/sec is logged.
Datasets are not shipped: ``data.class_name: synthetic`` draws U(-1,1) images (benchmarking only); ``data.npy`` may point
at a uint8 [N,32,32,3] .npy file; ``ddm.data.CIFAR10`` + ``img_folder`` reads the standard python batches.  Any other
data section raises instead of training on noise.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from adm_amd.ddm.utils import construct_class_by_name  # noqa: E402
from adm_amd.optim import BucketedGradReducer, FlatParams, FusedAdamWEMA, ema_decay_at, lr_lambda  # noqa: E402


class Cfg(dict):
    """dict with attribute access and .get(), standing in for fvcore's CfgNode."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = Cfg(v) if isinstance(v, dict) else v

    __getattr__ = dict.get


def parse_args():
    ap = argparse.ArgumentParser(description="training DDM (MI355X hot path)")
    ap.add_argument("--cfg", type=str, required=True)
    ap.add_argument("--max-steps", type=int, default=None, help="stop early (smoke runs)")
    args = ap.parse_args()
    with open(args.cfg) as f:
        args.cfg = yaml.load(f, Loader=yaml.SafeLoader)
    return args


def _load_cifar10_batches(folder):
    """uint8 [50000,32,32,3] from ``<folder>/cifar-10-batches-py/data_batch_{1..5}`` (the layout ddm.data.CIFAR10 reads,
    /root/reference/ddm/data.py:61-98).  The batches are pickled dicts of numpy arrays and python lists; the unpickler below
    resolves nothing except numpy's array reconstructors."""
    import pickle

    class _NumpyOnly(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split(".")[0] == "numpy" and name in ("_reconstruct", "ndarray", "dtype", "scalar", "_frombuffer"):
                return super().find_class(module, name)
            if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name == "_reconstruct":
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a dataset file")

    base = os.path.join(folder, "cifar-10-batches-py")
    parts = []
    for i in range(1, 6):
        path = os.path.join(base, f"data_batch_{i}")
        if not os.path.exists(path):
            raise FileNotFoundError(f"data.img_folder: {path} does not exist")
        with open(path, "rb") as f:
            parts.append(np.asarray(_NumpyOnly(f, encoding="latin1").load()["data"], dtype=np.uint8))
    return np.vstack(parts).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)


class ImageStream:
    """Infinite iterator of {'image': [B,3,H,W] in [-1,1]} on the GPU (the batch-dict shape of ddm.data.CIFAR10).

    Sources, in this order: ``data.npy`` (uint8 [N,H,W,3]; must exist and match ``image_size``); ``data.class_name:
    ddm.data.CIFAR10`` + ``img_folder`` (the reference YAML, /root/reference/configs/cifar10/*.yaml); ``data.class_name:
    synthetic`` = U(-1,1) images.  Anything else raises: a config that names a real dataset never trains on noise silently."""

    def __init__(self, data_cfg, batch, image_size, device, seed):
        self.batch, self.size, self.device = batch, tuple(image_size), device
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.images = None
        data_cfg = data_cfg or {}
        path, cls = data_cfg.get("npy"), data_cfg.get("class_name")
        self.flip = bool(data_cfg.get("augment_horizontal_flip", True))
        arr = None
        if path:
            if not os.path.exists(path):
                raise FileNotFoundError(f"data.npy: {path} does not exist")
            arr = np.load(path, allow_pickle=False)
        elif cls in ("ddm.data.CIFAR10", "CIFAR10"):
            if not data_cfg.get("img_folder"):
                raise ValueError("data.class_name ddm.data.CIFAR10 needs data.img_folder")
            arr = _load_cifar10_batches(data_cfg.get("img_folder"))
        elif cls != "synthetic":
            raise NotImplementedError(f"data.class_name {cls!r}: only ddm.data.CIFAR10, a uint8 data.npy, or 'synthetic' "
                                      "(U(-1,1) images, benchmarking only) are implemented")
        if arr is not None:
            if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[-1] != 3 or tuple(arr.shape[1:3]) != self.size:
                raise ValueError(f"image array must be uint8 [N,{self.size[0]},{self.size[1]},3], got {arr.dtype} {arr.shape}")
            self.images = torch.from_numpy(np.ascontiguousarray(arr)).to(device).permute(0, 3, 1, 2).float() / 127.5 - 1.0

    def __iter__(self):
        return self

    def __next__(self):
        if self.images is None:
            return {"image": torch.rand(self.batch, 3, *self.size, device=self.device, generator=self.gen) * 2 - 1}
        idx = torch.randint(0, self.images.shape[0], (self.batch,), device=self.device, generator=self.gen)
        x = self.images[idx]
        if not self.flip:
            return {"image": x}
        flip = torch.rand(self.batch, device=self.device, generator=self.gen) < 0.5
        return {"image": torch.where(flip[:, None, None, None], x.flip(-1), x)}


def save_grid(img, path, nrow):
    from PIL import Image
    img = (img.clamp(0, 1) * 255).round().to(torch.uint8).cpu()
    B, C, H, W = img.shape
    rows = (B + nrow - 1) // nrow
    canvas = torch.zeros(C, rows * H, nrow * W, dtype=torch.uint8)
    for i in range(B):
        r, c = divmod(i, nrow)
        canvas[:, r * H:(r + 1) * H, c * W:(c + 1) * W] = img[i]
    Image.fromarray(canvas.permute(1, 2, 0).numpy()).save(path)


class Trainer:
    def __init__(self, model, stream, cfg, device, rank, world):
        t = cfg.trainer
        self.model, self.stream, self.cfg, self.device, self.rank, self.world = model, stream, cfg, device, rank, world
        self.accum = t.get("gradient_accumulate_every", 1)
        self.lr, self.min_lr = float(t.lr), float(t.get("min_lr", 0.0))
        self.train_num_steps = int(t.train_num_steps)
        self.save_every = int(t.get("save_and_sample_every", 10000))
        self.log_freq = int(t.get("log_freq", 500))
        self.ema_after, self.ema_every = int(t.get("ema_update_after_step", 10000)), int(t.get("ema_update_every", 8))
        self.results = t.results_folder
        self.flat = FlatParams(model)
        if world > 1:
            dist.broadcast(self.flat.flat, src=0)
        self.reducer = BucketedGradReducer(self.flat)
        self.opt = FusedAdamWEMA(self.flat, lr=self.lr, weight_decay=float(t.get("weight_decay", 1e-4)), max_norm=1.0,
                                 ema=(rank == 0))
        self.warmup_iter = int(t.get("warmup_iter", 5000))      # train_uncond_dpm.py:170
        self.step = 0
        self.ema_step = 0
        self.ema_initted = False
        if rank == 0:
            os.makedirs(self.results, exist_ok=True)
        milestone = t.get("resume_milestone", 0)
        if milestone and os.path.exists(os.path.join(self.results, f"model-{milestone}.pt")):
            self.load(milestone)

    def _lr_ratio(self, it):
        return lr_lambda(it, self.lr, self.min_lr, self.train_num_steps, self.warmup_iter)

    # ---- checkpoint layout of train_uncond_dpm.py:207-239 --------------------------------------
    def ema_state_dict(self):
        sd, names = {}, [n for n, p in self.model.named_parameters() if p.requires_grad]
        for n, p, o in zip(names, self.flat.params, self.flat.offsets):
            sd["online_model." + n] = p.detach().clone()
            sd["ema_model." + n] = self.opt.ema[o:o + p.numel()].view(p.shape).clone()
        for n, b in self.model.named_buffers():
            sd["online_model." + n] = b.clone()
            sd["ema_model." + n] = b.clone()
        sd["initted"] = torch.tensor([self.ema_initted])
        sd["step"] = torch.tensor([self.ema_step])
        return sd

    def save(self, milestone):
        if self.rank != 0:
            return
        data = {"step": self.step, "model": self.model.state_dict(),
                "opt": {"exp_avg": self.opt.m, "exp_avg_sq": self.opt.v, "step": self.opt.step_count},
                "lr_scheduler": {"last_epoch": self.step}, "ema": self.ema_state_dict(), "scaler": None}
        torch.save(data, os.path.join(self.results, f"model-{milestone}.pt"))

    def load(self, milestone):
        data = torch.load(os.path.join(self.results, f"model-{milestone}.pt"), map_location=self.device, weights_only=True)
        if "scale_factor" in data["model"] and hasattr(self.model, "scale_factor"):     # train_uncond_ldm.py:206-207
            self.model.scale_factor = data["model"]["scale_factor"].to(self.device)
        self.model.load_state_dict(data["model"])
        self.step = data["step"]
        opt = data.get("opt")
        if isinstance(opt, dict) and "exp_avg" in opt:                   # this build's flat layout
            self.opt.m.copy_(opt["exp_avg"]); self.opt.v.copy_(opt["exp_avg_sq"])
            self.opt.step_count = int(opt["step"])
        elif isinstance(opt, dict) and "state" in opt and "param_groups" in opt:
            # torch.optim.AdamW.state_dict() as the reference's Trainer.save writes it (train_uncond_dpm.py:178-179, 214):
            # per-parameter {'step','exp_avg','exp_avg_sq'} indexed in the order of filter(requires_grad, model.parameters()),
            # which is the order of the flat buffers
            state = opt["state"]
            if len(state) not in (0, len(self.flat.params)):
                raise RuntimeError(f"checkpoint optimiser state has {len(state)} parameters, the model {len(self.flat.params)}")
            steps = set()
            for idx, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
                st = state.get(idx)
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise RuntimeError(f"optimiser state {idx}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
                self.opt.m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                self.opt.v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            if len(steps) > 1:
                raise RuntimeError(f"per-parameter optimiser step counts differ ({sorted(steps)}): not representable in the fused step")
            if steps:
                self.opt.step_count = steps.pop()
        elif opt is not None:
            raise RuntimeError("checkpoint 'opt' entry has an unknown layout")
        if self.rank == 0 and "ema" in data:
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
            for n, p, o in zip(names, self.flat.params, self.flat.offsets):
                if "ema_model." + n in data["ema"]:
                    self.opt.ema[o:o + p.numel()].copy_(data["ema"]["ema_model." + n].reshape(-1))
            self.ema_step = int(data["ema"].get("step", torch.tensor([0]))[0])
            self.ema_initted = bool(data["ema"].get("initted", torch.tensor([False]))[0])
        from adm_amd import ops
        ops.invalidate_packed()

    def train(self, max_steps=None):
        last, seen = time.time(), 0
        end = self.train_num_steps if max_steps is None else min(self.train_num_steps, self.step + max_steps)
        while self.step < end:
            self.flat.zero_grad()
            loss_acc, log_acc = 0.0, {}
            for ga in range(self.accum):
                self.reducer.enabled = ga == self.accum - 1          # communicate on the last micro-step only
                batch = next(self.stream)
                if self.step == 0 and ga == 0 and hasattr(self.model, "on_train_batch_start"):
                    self.model.on_train_batch_start(batch)       # latent models: std-rescaling from the first batch
                    if self.world > 1 and isinstance(getattr(self.model, "scale_factor", None), torch.Tensor):
                        dist.broadcast(self.model.scale_factor, src=0)      # one factor for all ranks
                loss, log = self.model.training_step(batch)
                (loss / self.accum).backward()
                loss_acc += float(loss.detach()) / self.accum
                for k, v in log.items():
                    log_acc[k] = log_acc.get(k, 0.0) + float(v) / self.accum
                seen += batch["image"].shape[0] * self.world
            self.reducer.finish()
            # EMA.update (ddm/ema.py:153-170): every `update_every` calls; copy until update_after_step
            decay = None
            if self.rank == 0 and self.ema_step % self.ema_every == 0:
                if self.ema_step <= self.ema_after:
                    decay = 0.0                      # copy_params_from_model_to_ema
                elif not self.ema_initted:           # first update after the warm-up: hard copy (ema.py:153-156, initted False)
                    decay, self.ema_initted = 0.0, True
                else:
                    decay = ema_decay_at(self.ema_step + 1, update_after_step=self.ema_after)
            self.ema_step += 1
            self.opt.step(lr=self.lr * self._lr_ratio(self.step), grad_scale=1.0 / self.world, ema_decay=decay)
            self.step += 1
            if self.rank == 0 and (self.step % self.log_freq == 0 or self.step == end):
                dt = time.time() - last
                print(f"[Train Step] {self.step}/{self.train_num_steps}: loss={loss_acc:.4f} "
                      f"loss_simple={log_acc.get('train/loss_simple', 0):.5f} lr={self.lr * self._lr_ratio(self.step):.3e} "
                      f"grad_norm={self.opt.grad_norm(1.0 / self.world):.3f} images/sec={seen / dt:.1f}", flush=True)
                last, seen = time.time(), 0
            if self.step % self.save_every == 0:
                milestone = self.step // self.save_every
                self.save(milestone)
                if self.rank == 0:
                    self.model.eval()
                    img = self.model.sample(batch_size=16)
                    self.model.train()
                    save_grid(img, os.path.join(self.results, f"sample-{milestone}.png"), 4)
        if self.rank == 0:
            print("training complete")


def build_model(model_cfg):
    """unet (+ frozen first stage when the YAML has model.first_stage) + diffusion wrapper, by dotted class name, as
    /root/reference/train_uncond_dpm.py:40-46 and train_uncond_ldm.py:42-59 do."""
    unet = construct_class_by_name(**{k: v for k, v in model_cfg.unet.items()})
    kw = {k: v for k, v in model_cfg.items() if k not in ("class_name", "unet", "first_stage")}
    if model_cfg.get("first_stage"):
        kw["auto_encoder"] = construct_class_by_name(**{k: v for k, v in model_cfg.first_stage.items()})
    return construct_class_by_name(model=unet, cfg=model_cfg, class_name=model_cfg.class_name, **kw)


def main(args):
    cfg = Cfg(args.cfg)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("ADM_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ADM_DIST_BACKEND", "nccl")       # "nccl" = RCCL; gloo only for one-card rehearsals in tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    model_cfg = cfg.model
    dpm = build_model(model_cfg).to(device).train()
    global_batch = int(cfg.data.batch_size)
    assert global_batch % world == 0, "split_batches: the YAML batch_size is the global batch"
    stream = ImageStream(cfg.data, global_batch // world, tuple(cfg.data.get("image_size") or model_cfg.image_size), device,
                         seed=1000 + rank)
    trainer = Trainer(dpm, stream, cfg, device, rank, world)
    if cfg.trainer.get("test_before", False) and rank == 0:
        dpm.eval()
        save_grid(dpm.sample(batch_size=16), os.path.join(cfg.trainer.results_folder, "sample-0.png"), 4)
        dpm.train()
    trainer.train(args.max_steps)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
