#!/usr/bin/env python3
"""Count the ATen ops (and where they are called from) in one training step of the bench model: everything that is
not one of our C-ABI kernels shows up here (autograd's gradient sums, stray copies, ...).  Debug tool; GPU box only."""
import os
import sys
from collections import Counter

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
from adm_amd.optim import FlatParams, FusedAdamWEMA  # noqa: E402

dpm = bench.build_model(dev, small="--small" in sys.argv).train()
flat = FlatParams(dpm)
opt = FusedAdamWEMA(flat, lr=1e-4, weight_decay=1e-4, max_norm=1.0, ema=True)
B = 128
batch = {"image": torch.rand(B, 3, 32, 32, device=dev) * 2 - 1}


def step():
    flat.zero_grad()
    loss, _ = dpm.training_step(batch)
    loss.backward()
    opt.step(lr=1e-4, grad_scale=1.0, ema_decay=0.999)


step(); step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt, where = Counter(), {}
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::add", "aten::add_", "aten::clone", "aten::contiguous",
                                                   "aten::mul", "aten::zeros", "aten::zero_", "aten::fill_", "aten::sum",
                                                   "aten::cat", "aten::stack", "aten::empty_like", "aten::to", "aten::_to_copy"):
        shp = str(e.input_shapes)[:60]
        cnt[(e.name, shp)] += 1
        if (e.name, shp) not in where:
            where[(e.name, shp)] = [s for s in e.stack if "adm_amd" in s or "bench" in s or "autograd" in s][:3]
for (name, shp), n in cnt.most_common(40):
    print(f"{n:5d} {name:18s} {shp:62s} {where[(name, shp)]}")
