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
Datasets are not shipped: ``data.class_name: synthetic`` (default when the folder is missing) draws
U(-1,1) images; ``data.npy`` may point at a uint8 [N,32,32,3] .npy file.
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


class ImageStream:
    """Infinite iterator of {'image': [B,3,H,W] in [-1,1]} on the GPU (the batch-dict shape of ddm.data.CIFAR10)."""

    def __init__(self, data_cfg, batch, image_size, device, seed):
        self.batch, self.size, self.device = batch, image_size, device
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.images = None
        path = data_cfg.get("npy") if data_cfg else None
        if path and os.path.exists(path):
            arr = np.load(path, allow_pickle=False)          # uint8 [N,H,W,3]
            self.images = torch.from_numpy(arr).to(device).permute(0, 3, 1, 2).float() / 127.5 - 1.0

    def __next__(self):
        if self.images is None:
            return {"image": torch.rand(self.batch, 3, *self.size, device=self.device, generator=self.gen) * 2 - 1}
        idx = torch.randint(0, self.images.shape[0], (self.batch,), device=self.device, generator=self.gen)
        x = self.images[idx]
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
        self.step = 0
        self.ema_step = 0
        if rank == 0:
            os.makedirs(self.results, exist_ok=True)
        milestone = t.get("resume_milestone", 0)
        if milestone and os.path.exists(os.path.join(self.results, f"model-{milestone}.pt")):
            self.load(milestone)

    # ---- checkpoint layout of train_uncond_dpm.py:207-239 --------------------------------------
    def ema_state_dict(self):
        sd, names = {}, [n for n, p in self.model.named_parameters() if p.requires_grad]
        for n, p, o in zip(names, self.flat.params, self.flat.offsets):
            sd["online_model." + n] = p.detach().clone()
            sd["ema_model." + n] = self.opt.ema[o:o + p.numel()].view(p.shape).clone()
        for n, b in self.model.named_buffers():
            sd["online_model." + n] = b.clone()
            sd["ema_model." + n] = b.clone()
        sd["initted"] = torch.tensor([self.step > self.ema_after])
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
        if isinstance(data.get("opt"), dict) and "exp_avg" in data["opt"]:
            self.opt.m.copy_(data["opt"]["exp_avg"]); self.opt.v.copy_(data["opt"]["exp_avg_sq"])
            self.opt.step_count = int(data["opt"]["step"])
        if self.rank == 0 and "ema" in data:
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
            for n, p, o in zip(names, self.flat.params, self.flat.offsets):
                if "ema_model." + n in data["ema"]:
                    self.opt.ema[o:o + p.numel()].copy_(data["ema"]["ema_model." + n].reshape(-1))
            self.ema_step = int(data["ema"].get("step", torch.tensor([0]))[0])
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
                decay = 0.0 if self.ema_step <= self.ema_after else ema_decay_at(self.ema_step + 1, update_after_step=self.ema_after)
            self.ema_step += 1
            self.opt.step(lr=self.lr * lr_lambda(self.step, self.lr, self.min_lr, self.train_num_steps),
                          grad_scale=1.0 / self.world, ema_decay=decay)
            self.step += 1
            if self.rank == 0 and (self.step % self.log_freq == 0 or self.step == end):
                dt = time.time() - last
                print(f"[Train Step] {self.step}/{self.train_num_steps}: loss={loss_acc:.4f} "
                      f"loss_simple={log_acc.get('train/loss_simple', 0):.5f} lr={self.lr * lr_lambda(self.step, self.lr, self.min_lr, self.train_num_steps):.3e} "
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
