#!/usr/bin/env python3
"""GroupNorm(+scale/shift+SiLU+dropout) forward / backward timing per shape, one-launch path on and off (diagnostic; GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adm_amd import hip, ops  # noqa: E402

B = 128
dev = torch.device("cuda:0")
lib = hip.lib()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for HW, C in [(1024, 192), (1024, 384), (1024, 576), (256, 192), (256, 384), (256, 576), (256, 768), (64, 384), (64, 768), (16, 384)]:
    h = int(HW ** 0.5)
    x = torch.randn(B, h, h, C, device=dev)
    gam, bet = torch.randn(C, device=dev).requires_grad_(True), torch.randn(C, device=dev).requires_grad_(True)
    ss = torch.randn(B, 2 * C, device=dev) * 0.1
    gy = torch.randn(B, h, h, C, device=dev)
    line = f"HW={HW:5d} C={C:4d}:"
    for fused in (0, 1):
        lib.adm_gn_fused(fused)
        xr = x.clone().requires_grad_(True)
        tf = timeit(lambda: ops.group_norm_act(xr.detach(), gam.detach(), bet.detach(), ss, silu=True, drop_p=0.1))
        y = ops.group_norm_act(xr, gam, bet, ss, silu=True, drop_p=0.1)
        tb = timeit(lambda: torch.autograd.grad(y, (xr, gam, bet), gy, retain_graph=True))
        n = B * HW * C
        line += f"  fused={fused}: fwd {tf * 1e3:7.1f} us ({8 * n / tf / 1e9:5.2f} TB/s min)  bwd {tb * 1e3:7.1f} us ({12 * n / tb / 1e9:5.2f} TB/s min)"
    lib.adm_gn_fused(1)
    print(line, flush=True)
