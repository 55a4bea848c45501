#!/usr/bin/env python3
"""Which host-side ops issue the small device copies / fills of one training step (diagnostic; GPU box)."""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from adm_amd.optim import FlatParams, FusedAdamWEMA  # noqa: E402

dev = torch.device("cuda:0")
dpm = bench.build_model(dev, small=False).train()
flat = FlatParams(dpm)
opt = FusedAdamWEMA(flat, lr=1e-4, weight_decay=1e-4, max_norm=1.0, ema=True)
batch = {"image": torch.rand(128, 3, 32, 32, device=dev) * 2 - 1}


def step():
    flat.zero_grad()
    loss, _ = dpm.training_step(batch)
    loss.backward()
    opt.step(lr=1e-4, grad_scale=1.0, ema_decay=0.999)


step(); step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = Counter()
for e in prof.events():
    n = e.name
    if ("emcpy" in n or "emset" in n) and e.device_type == torch.autograd.DeviceType.CPU:
        par, chain = e.cpu_parent, []
        while par is not None and len(chain) < 4:
            chain.append(par.name[:48]); par = par.cpu_parent
        cnt[(n[:28], " < ".join(chain))] += 1
for (n, chain), k in cnt.most_common(30):
    print(f"{k:5d} {n:28s} {chain}")
