"""GPU: ADM_DETERMINISTIC=1 (adm_amd.ops.DETERMINISTIC) -- SURVEY.md section 5.2 "run twice, bit-compare".
The weight / bias gradient kernels' pixel-range splits go to a workspace and are summed in a fixed order instead of with
fp32 atomics; the gradient norm uses per-workgroup partials.  Checked: (1) the deterministic path still matches
F.conv2d's autograd, (2) it is bitwise reproducible where the atomic path need not be, (3) two complete training steps
(loss, every gradient, every updated parameter and optimiser moment) are bit-identical."""
import itertools
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import fill, unet_ref
from parity import close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import hip, ops as _ops
    hip.lib()
    monkeypatch.setattr(_ops, "DETERMINISTIC", True)
    return _ops


@pytest.mark.parametrize("B,cin,cout,H,ks,up,qkv", [(8, 64, 96, 32, 3, False, False), (4, 96, 64, 16, 3, True, False), (32, 64, 64, 16, 3, False, False),
                                                    (16, 128, 64, 8, 3, False, False), (8, 64, 192, 16, 1, False, True),
                                                    (3, 32, 3, 12, 3, False, False), (5, 768, 384, 1, 1, False, False)])
def test_deterministic_weight_gradient_matches_and_repeats(ops, B, cin, cout, H, ks, up, qkv):
    x = fill.hash_tensor((B, cin, H, H), f"dx{cin}{cout}{H}", 1.0)
    w = fill.hash_tensor((cout, cin, ks, ks), f"dw{cin}{cout}", 1.0 / math.sqrt(cin * ks * ks))
    b = fill.hash_tensor((cout,), f"db{cin}{cout}", 0.5)
    Ho = 2 * H if up else H
    gy = fill.hash_tensor((B, cout, Ho, Ho), f"dg{cin}{cout}{H}", 1.0)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    (F.conv2d(xin, wr, br, padding=ks // 2) * gy).sum().backward()
    cip, cop = ops.ceil32(cin), ops.ceil32(cout)
    pad = lambda t, c: torch.cat([t, torch.zeros(t.shape[0], c - t.shape[1], *t.shape[2:])], 1) if c > t.shape[1] else t
    xd = pad(x, cip).permute(0, 2, 3, 1).contiguous().cuda()
    gyd = pad(gy, cop).permute(0, 2, 3, 1).contiguous().cuda()
    if qkv:       # packed output-channel order (head, {q,k,v}, c): weight the outputs accordingly
        heads = cout // 192
        gyd = gyd.reshape(B, Ho, Ho, heads, 64, 3).permute(0, 1, 2, 3, 5, 4).reshape(B, Ho, Ho, cout).contiguous()
    runs = []
    for _ in range(2):
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        y = ops.conv2d(xd, wd, bd, None, up=up, qkv=qkv)
        (y * gyd).sum().backward()
        runs.append((wd.grad.clone(), bd.grad.clone()))
    close(runs[0][0], wr.grad)
    close(runs[0][1], br.grad)
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def test_two_training_steps_are_bitwise_identical(ops):
    """Two optimiser steps of the reduced two-decoder model at batch 8 (32x32 layers: M = 8192, so the Winograd kernels and
    multi-split weight gradients are the ones that run), dropout 0.1, clip + AdamW + EMA: run the whole thing twice from
    the same state and compare every bit of the losses, the flat gradient buffer and the optimiser state."""
    from adm_amd.ddm.ddm_const import DDPM
    from adm_amd.optim import FlatParams, FusedAdamWEMA
    from adm_amd.unet.uncond_unet import EDMPrecond
    cfg = unet_ref.default_cfg(variant="uncond_unet", model_channels=64, num_blocks=1, dropout=0.1)
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions",
                              "dropout", "augment_dim")}
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    x = (fill.hash_tensor((8, 3, 32, 32), "det.x", 1.0)).cuda()

    def run():
        torch.manual_seed(11)
        ops._drop_counter = itertools.count(1)
        unet = EDMPrecond(img_resolution=32, img_channels=3, model_type="DhariwalUNet", **kw)
        unet.load_state_dict(sd, strict=True)
        dpm = DDPM(model=unet, image_size=[32, 32], sampling_timesteps=10, perceptual_weight=0.0,
                   cfg=dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True)).cuda().train()
        flat = FlatParams(dpm)
        opt = FusedAdamWEMA(flat, lr=1e-3, weight_decay=1e-4, max_norm=1.0, ema=True)
        out = []
        for it in range(2):
            flat.zero_grad()
            loss, _ = dpm.training_step({"image": x})
            loss.backward()
            out += [loss.detach().clone(), flat.grad.clone()]
            opt.step(lr=1e-3, ema_decay=0.9 if it else 0.0)
            out += [opt.sumsq.clone()]
        return out + [flat.flat.clone(), opt.m.clone(), opt.v.clone(), opt.ema.clone()]

    a, b = run(), run()
    assert float(a[1].abs().max()) > 0
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), f"output {i} differs between two identical runs (max diff {float((u - v).abs().max()):.3e})"
