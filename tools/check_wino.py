#!/usr/bin/env python3
"""Winograd F(2,3) conv kernel vs the direct implicit GEMM and vs torch, plus timing (GPU box only)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adm_amd import hip, ops  # noqa: E402
from adm_amd.hip import call, ptr  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = ((2, 8, 8, 32, 32), (3, 16, 8, 64, 96), (1, 4, 6, 48, 32), (128, 32, 32, 384, 384), (128, 16, 16, 768, 384),
          (128, 32, 32, 192, 192), (128, 8, 8, 384, 384))
if len(sys.argv) > 1:
    SHAPES = tuple(tuple(int(v) for v in a.split(',')) for a in sys.argv[1:])
for (B, H, W, ci, co) in SHAPES:
    x = torch.randn(B, H, W, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * (1.0 / (9 * ci)) ** 0.5
    b = torch.randn(co, device=dev)
    res = torch.randn(B, H, W, co, device=dev)
    cip, cop = ops.ceil32(ci), ops.ceil32(co)
    xp = torch.zeros(B, H, W, cip, device=dev); xp[..., :ci] = x
    wf = torch.empty(4, cop, 3, cip, device=dev)
    wb = torch.empty(4, cip, 3, cop, device=dev)
    call("adm_pack_weight_wino", ptr(w), ptr(wf), ptr(wb), co, ci, cop, cip)
    y = torch.empty(B, H, W, cop, device=dev)
    bp = torch.zeros(cop, device=dev); bp[:co] = b
    rp = torch.zeros(B, H, W, cop, device=dev); rp[..., :co] = res
    call("adm_conv_fwd_wino", ptr(xp), ptr(wf), ptr(bp), ptr(rp), ptr(y), B, H, W, cip, cip, cop, cop, cop, cop)
    with torch.no_grad():
        yd = ops.conv2d(xp, w, b, rp)
    err_d = float((y - yd).abs().max() / yd.abs().max())
    msg = f"B={B} H={H} W={W} ci={ci} co={co}: wino vs direct rel {err_d:.2e}"
    if B * H * W <= 4096:
        ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1) + res.double().cpu()
        msg += f"; vs fp64 torch: wino {float((y[..., :co].cpu() - ref).abs().max() / ref.abs().max()):.2e} direct {float((yd[..., :co].cpu() - ref).abs().max() / ref.abs().max()):.2e}"
        # data gradient: dx = conv(dy, flipped/transposed weights)
        dy = torch.randn(B, H, W, cop, device=dev); dy[..., co:] = 0
        dx = torch.empty(B, H, W, cip, device=dev)
        call("adm_conv_fwd_wino", ptr(dy), ptr(wb), None, None, ptr(dx), B, H, W, cop, cop, cip, cip, cip, cip)
        xr = x.permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
        (F.conv2d(xr, w.double().cpu(), None, padding=1) * dy[..., :co].permute(0, 3, 1, 2).double().cpu()).sum().backward()
        msg += f"; dgrad rel {float((dx[..., :ci].cpu().permute(0, 3, 1, 2) - xr.grad).abs().max() / xr.grad.abs().max()):.2e}"
    else:
        for name, fn in (("wino", lambda: call("adm_conv_fwd_wino", ptr(xp), ptr(wf), ptr(bp), ptr(rp), ptr(y), B, H, W, cip, cip, cop, cop, cop, cop)),
                         ("direct", lambda: call("adm_conv_fwd", ptr(xp), ptr(w._adm_packed.fwd), ptr(bp), ptr(rp), ptr(yd), B, H, W, cip, cip, cop, cop, cop, cop, 3, 0, -1))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            msg += f"; {name} {dt * 1e3:.3f} ms = {2.0 * B * H * W * ci * co * 9 / dt / 1e12:.1f} TF(alg)"
    print(msg, flush=True)
