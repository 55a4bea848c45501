#!/usr/bin/env python3
"""MI355X counterpart of the reference's unconditional sampler (/root/reference/sample_uncond.py).

``python sample_uncond.py --cfg <yaml>``: builds unet + DDPM from the YAML exactly as the trainer does,
loads ``cfg.sampler.ckpt_path`` (taking the EMA weights when ``sampler.use_ema``: the checkpoint's 'ema'
dict with the ``ema_model.`` prefix stripped, reference :131-147), then draws ``sampler.sample_num`` images
in batches of ``sampler.batch_size`` with ``model.sample(batch_size=...)`` (10-step deterministic sampler)
and writes them as PNGs named ``f'{i: 010d}.png'`` (reference :168-173 -- note the space flag).
Each rank samples its own disjoint id range with its own seed; there are no collectives (the reference
lets every rank write the same names).
"""
import argparse
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from train_uncond_dpm import Cfg, build_model  # noqa: E402


def load_weights(model, path, use_ema, device):
    data = torch.load(path, map_location=device, weights_only=True)
    if use_ema and "ema" in data:
        sd = {k[len("ema_model."):]: v for k, v in data["ema"].items() if k.startswith("ema_model.")}
    else:
        sd = data["model"] if "model" in data else data
    if "model" in data and "scale_factor" in data["model"] and hasattr(model, "scale_factor"):     # reference :146-147
        model.scale_factor = data["model"]["scale_factor"].to(device)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    matched = len(sd) - len(unexpected)
    print(f"loaded {path}: {matched} tensors matched, {len(missing)} missing, {len(unexpected)} unexpected keys")
    if matched == 0:
        raise RuntimeError(f"{path}: none of the checkpoint's {len(sd)} keys matches the model (wrong config or prefix?)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", required=True)
    ap.add_argument("--max-batches", type=int, default=None)
    ap.add_argument("--random-init", action="store_true", help="smoke runs: sample without loading a checkpoint")
    args = ap.parse_args()
    with open(args.cfg) as f:
        cfg = Cfg(yaml.load(f, Loader=yaml.SafeLoader))
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.manual_seed(42 + rank)
    mc = cfg.model
    dpm = build_model(mc).to(device).eval()          # pixel-space DDPM or LatentDiffusion (+ first stage), reference :50-64
    s = cfg.sampler
    if s.get("ckpt_path") and not args.random_init:
        if not os.path.exists(s.ckpt_path):
            raise FileNotFoundError(f"sampler.ckpt_path {s.ckpt_path} does not exist (pass --random-init to sample from "
                                    "freshly initialised weights)")
        load_weights(dpm, s.ckpt_path, s.get("use_ema", True), device)
    elif not args.random_init:
        raise ValueError("sampler.ckpt_path is empty (pass --random-init to sample from freshly initialised weights)")
    else:
        print("--random-init: sampling from the initialised weights")
    out = s.save_folder
    os.makedirs(out, exist_ok=True)
    from PIL import Image
    per_rank = int(s.sample_num) // world
    bs = int(s.batch_size)
    n_batches = (per_rank + bs - 1) // bs
    if args.max_batches is not None:
        n_batches = min(n_batches, args.max_batches)
    img_id, t0, done = rank * per_rank, time.time(), 0
    for _ in range(n_batches):
        real_bs = min(bs, per_rank - done)
        batch = dpm.sample(batch_size=real_bs)
        arr = (batch.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        for j in range(real_bs):
            Image.fromarray(arr[j]).save(os.path.join(out, f"{img_id: 010d}.png"))
            img_id += 1
        done += real_bs
    torch.cuda.synchronize()
    print(f"rank {rank}: {done} images in {time.time() - t0:.1f}s ({done / (time.time() - t0):.1f} images/sec incl. PNG writes)")


if __name__ == "__main__":
    main()
