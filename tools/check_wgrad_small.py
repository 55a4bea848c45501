#!/usr/bin/env python3
"""Weight-gradient kernels on the small feature maps: direct vs Winograd (GPU box only)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adm_amd import ops  # noqa: E402
from adm_amd.hip import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
for (B, H, ci, co) in ((128, 4, 384, 384), (128, 4, 768, 384), (128, 8, 384, 384), (128, 8, 768, 384)):
    x = torch.randn(B, H, H, ci, device=dev)
    dy = torch.randn(B, H, H, co, device=dev)
    dwp = torch.empty(co, 9, ci, device=dev)
    out = {}
    for name, fn in (("direct", lambda: call("adm_conv_wgrad_bias", ptr(x), ptr(dy), ptr(dwp), None, B, H, H, ci, ci, co, co, 3, 0, 0)),
                     ("wino", lambda: call("adm_conv_wgrad_wino", ptr(x), ptr(dy), ptr(dwp), None, B, H, H, ci, ci, co, co, 0))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        out[name] = (dt, dwp.clone())
    err = float((out["wino"][1] - out["direct"][1]).abs().max() / out["direct"][1].abs().max())
    fl = 2.0 * B * H * H * ci * co * 9
    print(f"B={B} H={H} ci={ci} co={co}: direct {out['direct'][0] * 1e6:.1f} us ({fl / out['direct'][0] / 1e12:.1f} TF)  "
          f"wino {out['wino'][0] * 1e6:.1f} us ({fl / out['wino'][0] / 1e12:.1f} TF)  rel diff {err:.1e}", flush=True)
