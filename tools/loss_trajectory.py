#!/usr/bin/env python3
"""Loss trajectory of the full-size bench model over a few optimiser steps on a fixed synthetic batch stream (fixed seeds): run it
under ADM_FP16X3=1 / 0 (or ADM_BF16X6=0) and compare the printed losses -- the number formats of the split kernels must not
change the training dynamics beyond f32 rounding.  GPU box only."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from adm_amd.optim import FlatParams, FusedAdamWEMA  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
torch.manual_seed(0)
dpm = bench.build_model(dev).train()
flat = FlatParams(dpm)
opt = FusedAdamWEMA(flat, lr=1e-4, weight_decay=1e-4, max_norm=1.0, ema=True)
g = torch.Generator(device="cpu").manual_seed(1)
losses = []
for it in range(steps):
    batch = {"image": (torch.rand(128, 3, 32, 32, generator=g) * 2 - 1).to(dev)}
    t = (torch.rand(128, generator=g) * 0.999 + 0.001).to(dev)
    noise = torch.randn(128, 3, 32, 32, generator=g).to(dev)
    flat.zero_grad()
    loss, _ = dpm.training_step(batch, t=t, noise=noise)
    loss.backward()
    opt.step(lr=1e-4, grad_scale=1.0, ema_decay=0.999)
    losses.append(float(loss.detach()))
print("losses " + " ".join(f"{v:.6f}" for v in losses))
