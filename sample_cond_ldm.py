#!/usr/bin/env python3
"""MI355X counterpart of the reference's conditional latent sampler (/root/reference/sample_cond_ldm.py): super-resolution by
SLIDING-WINDOW sampling with overlap averaging (reference :281-330, `slide_sample_sr`).

``python sample_cond_ldm.py --cfg <yaml>`` builds unet (unet.cond_unet.Unet / unet.cond_unet_sd.Unet) + frozen first stage +
ddm.ddm_const.LatentDiffusion from the YAML exactly as the reference does (:53-66), loads ``sampler.ckpt_path`` (EMA weights
when ``sampler.use_ema``, prefix stripped, :135-147), then for every low-resolution condition image cuts windows of
``sampler.crop_size`` at ``sampler.stride`` (origins clamped to the border, :296-301), samples each window's 4x larger
output crop with the ``sampling_timesteps``-step sampler conditioned on the window, and averages overlapping outputs.

What differs from the reference:
  * windows are sampled in BATCHES (``sampler.window_batch``, default all windows of an image at once) instead of one
    ``model.sample(batch_size=1)`` call per window: each window's trajectory is independent, so batching changes only which
    random draws a window receives, and keeps the GPU full; ``window_batch: 1`` reproduces the reference's call pattern;
  * the condition encoder (torchvision Swin-B + fetched ImageNet weights in the reference) is NOT part of this build:
    ``sampler.cond_encoder`` names a callable ``module:attr`` that maps a condition crop [B,3,h,w] to the four feature maps
    the denoiser consumes; ``synthetic`` selects a parameter-free pooled pyramid (smoke runs / benchmarking only);
  * every rank handles a disjoint share of the images; no collectives.
"""
import argparse
import importlib
import math
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from train_uncond_dpm import Cfg, build_model  # noqa: E402
from sample_uncond import load_weights  # noqa: E402


def slide_windows(h_cond, w_cond, crop, stride):
    """(y1, y2, x1, x2) of every window, in the reference's order (sample_cond_ldm.py:288-301)."""
    (hc, wc), (hs, ws) = crop, stride
    hg = max(h_cond - hc + hs - 1, 0) // hs + 1
    wg = max(w_cond - wc + ws - 1, 0) // ws + 1
    out = []
    for hi in range(hg):
        for wi in range(wg):
            y2 = min(hi * hs + hc, h_cond); x2 = min(wi * ws + wc, w_cond)
            out.append((max(y2 - hc, 0), y2, max(x2 - wc, 0), x2))
    return out


@torch.no_grad()
def slide_sample_sr(sample_fn, cond, image_hw, crop_size, stride, out_channels=3, scale=4, ori_size=None, window_batch=0,
                    flip_test=False):
    """cond [B,C,h,w] -> [B,out_channels,H,W]: mean over the windows covering each output pixel of sample_fn(window) (the 4x
    larger crop).  Windows of equal size are stacked along the batch axis, `window_batch` at a time (0 = all)."""
    B = cond.shape[0]
    H, W = image_hw
    preds = torch.zeros((B, out_channels, H, W), device=cond.device, dtype=torch.float32)
    count = torch.zeros((1, 1, H, W), device=cond.device, dtype=torch.float32)
    wins = slide_windows(cond.shape[2], cond.shape[3], crop_size, stride)
    step = len(wins) if window_batch <= 0 else window_batch
    for i in range(0, len(wins), step):
        chunk = wins[i:i + step]
        crops = torch.cat([cond[:, :, y1:y2, x1:x2] for (y1, y2, x1, x2) in chunk], dim=0).contiguous()
        out = sample_fn(crops).to(torch.float32)
        if flip_test:          # reference :307-310: average with the sample of the mirrored condition, mirrored back
            out = 0.5 * out + 0.5 * sample_fn(crops.flip(dims=[-1])).to(torch.float32).flip(dims=[-1])
        for k, (y1, y2, x1, x2) in enumerate(chunk):
            preds[:, :, y1 * scale:y2 * scale, x1 * scale:x2 * scale] += out[k * B:(k + 1) * B]
            count[:, :, y1 * scale:y2 * scale, x1 * scale:x2 * scale] += 1
    assert int((count == 0).sum()) == 0
    res = preds / count
    return res if ori_size is None else res[:, :, :ori_size[0], :ori_size[1]]


class SyntheticCondEncoder(torch.nn.Module):
    """Parameter-free stand-in for the condition backbone's OUTPUT SHAPES (f, 2f, 4f, 8f channels at 1/4 ... 1/32): average-pooled
    copies of the condition image tiled across channels.  For smoke runs and throughput measurement only."""

    def __init__(self, f=128):
        super().__init__()
        self.f = f

    def forward(self, x):
        feats = []
        for i in range(4):
            p = torch.nn.functional.avg_pool2d(x.float(), 4 << i) if min(x.shape[-2:]) >= (4 << i) else x.float().mean((2, 3), keepdim=True)
            c = self.f << i
            feats.append(p.repeat(1, (c + p.shape[1] - 1) // p.shape[1], 1, 1)[:, :c].contiguous())
        return feats


def resolve_encoder(spec, f=128):
    if spec in (None, "", "none"):
        return None
    if spec == "synthetic":
        return SyntheticCondEncoder(f)
    mod, attr = spec.split(":")
    obj = getattr(importlib.import_module(mod), attr)
    return obj() if isinstance(obj, type) else obj


class CondStream:
    """Condition / target pairs.  ``data.class_name: synthetic`` draws U(-1,1) high-resolution images and box-downsamples them 4x
    for the condition (the geometry of ddm.data.SRDatasetTest: 'image', 'cond', 'ori_size', 'img_name'); ``data.npy`` may hold
    uint8 [N,H,W,3] high-resolution images.  Anything else raises (no silent noise for a config that names a real dataset)."""

    def __init__(self, data_cfg, n, device, seed):
        import numpy as np
        self.n, self.device = n, device
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.images = None
        self.size = tuple(data_cfg.get("image_size") or (512, 512))
        path, cls = data_cfg.get("npy"), data_cfg.get("class_name")
        if path:
            if not os.path.exists(path):
                raise FileNotFoundError(f"data.npy: {path} does not exist")
            arr = np.load(path, allow_pickle=False)
            if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[-1] != 3:
                raise ValueError(f"image array must be uint8 [N,H,W,3], got {arr.dtype} {arr.shape}")
            self.images = torch.from_numpy(arr).permute(0, 3, 1, 2).float() / 127.5 - 1.0
        elif cls != "synthetic":
            raise NotImplementedError(f"data.class_name {cls!r}: only a uint8 data.npy or 'synthetic' are implemented")

    def __iter__(self):
        for i in range(self.n):
            if self.images is not None:
                img = self.images[i % self.images.shape[0]][None].to(self.device)
            else:
                img = torch.rand(1, 3, *self.size, device=self.device, generator=self.gen) * 2 - 1
            H, W = img.shape[-2:]
            Hp, Wp = (H + 3) // 4 * 4, (W + 3) // 4 * 4
            pad = torch.nn.functional.pad(img, (0, Wp - W, 0, Hp - H), mode="replicate")
            yield {"image": pad, "cond": torch.nn.functional.avg_pool2d(pad, 4), "ori_size": (H, W), "img_name": f"{i: 010d}.png"}


def main():
    ap = argparse.ArgumentParser(description="sliding-window conditional latent sampler (MI355X hot path)")
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--max-images", type=int, default=None)
    ap.add_argument("--random-init", action="store_true", help="smoke runs: sample without loading a checkpoint")
    args = ap.parse_args()
    with open(args.cfg) as f:
        cfg = Cfg(yaml.load(f, Loader=yaml.SafeLoader))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.manual_seed(42 + rank)
    mc, s = cfg.model, cfg.sampler
    assert mc.get("ldm"), "this driver is for latent models (reference :58)"
    ldm = build_model(mc).to(device).eval()
    enc = resolve_encoder(s.get("cond_encoder"), getattr(ldm.model, "f_cond", 128))
    if enc is None:
        raise ValueError("sampler.cond_encoder is required: 'module:callable' returning the four condition feature maps, or "
                         "'synthetic' (the reference's torchvision Swin-B and its weights are not available offline)")
    ldm.model.init_conv_mask = enc.to(device) if isinstance(enc, torch.nn.Module) else enc
    if s.get("ckpt_path") and not args.random_init:
        if not os.path.exists(s.ckpt_path):
            raise FileNotFoundError(f"sampler.ckpt_path {s.ckpt_path} does not exist (pass --random-init for a smoke run)")
        load_weights(ldm, s.ckpt_path, s.get("use_ema", True), device)
    elif not args.random_init:
        raise ValueError("sampler.ckpt_path is empty (pass --random-init for a smoke run)")
    out_dir = s.save_folder
    os.makedirs(out_dir, exist_ok=True)
    from PIL import Image
    n_total = int(s.sample_num) if args.max_images is None else min(int(s.sample_num), args.max_images)
    per_rank = n_total // world
    stream = CondStream(cfg.data, per_rank, device, seed=2000 + rank)
    crop, stride = tuple(s.crop_size), tuple(s.stride)
    psnr, done, t0 = 0.0, 0, time.time()
    for batch in stream:
        cond, image = batch["cond"], (batch["image"] + 1) * 0.5
        down = ldm.first_stage_model.down_ratio
        # the encoder runs ONCE per window batch (the reference re-runs it inside every denoising step, cond_unet_sd.py:821)
        fn = lambda c: ldm.sample(cond=list(ldm.model.init_conv_mask(c)), latent_hw=(c.shape[2] * 4 // down, c.shape[3] * 4 // down))
        pred = slide_sample_sr(fn, cond, image.shape[-2:], crop, stride,
                               out_channels=int(s.get("out_channels", 3)), ori_size=batch["ori_size"],
                               window_batch=int(s.get("window_batch", 0)), flip_test=bool(s.get("flip_test", False)))
        H, W = batch["ori_size"]
        mse = torch.mean((pred - image[:, :, :H, :W]) ** 2)
        psnr += float(-10.0 * torch.log10(mse))
        arr = (pred[0].clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()
        name = batch["img_name"] if world == 1 else f"{rank * per_rank + done: 010d}.png"
        Image.fromarray(arr).save(os.path.join(out_dir, name))
        done += 1
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"rank {rank}: {done} images in {dt:.1f}s ({done / max(dt, 1e-9):.2f} images/sec incl. PNG writes); PSNR: {psnr / max(done, 1):.2f}")


if __name__ == "__main__":
    main()
