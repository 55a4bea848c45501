"""GPU: size-independent properties at BASELINE.json's full sizes (bs=128, full channel widths), where the
CPU oracle would take minutes: linearity of the convolution kernels, GroupNorm's affine invariance,
attention's key-permutation invariance, wgrad/dgrad adjointness, determinism, and the sampler's fixed
point."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import hip, ops as _ops
    hip.lib()
    return _ops


def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.rand(shape, device="cuda", generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("B,H,cin,cout,ks,up", [(128, 32, 192, 192, 3, False), (128, 16, 768, 384, 3, False),
                                                 (128, 16, 384, 384, 3, True), (128, 16, 384, 1152, 1, False),
                                                 (128, 4, 768, 384, 3, False)])
def test_conv_linearity_and_adjointness_full_size(ops, B, H, cin, cout, ks, up):
    x1, x2 = rnd((B, H, H, cin), 1), rnd((B, H, H, cin), 2)
    w = rnd((cout, cin, ks, ks), 3, 1 / math.sqrt(cin * ks * ks)).requires_grad_(True)
    b = rnd((cout,), 4)
    zero_b = torch.zeros_like(b)
    y1, y2 = ops.conv2d(x1, w.detach(), zero_b, up=up), ops.conv2d(x2, w.detach(), zero_b, up=up)
    y12 = ops.conv2d(0.7 * x1 - 1.3 * x2, w.detach(), zero_b, up=up)
    torch.testing.assert_close(y12, 0.7 * y1 - 1.3 * y2, rtol=1e-3, atol=1e-4)
    # bias and residual are pure additions
    r = rnd(y1.shape, 5)
    torch.testing.assert_close(ops.conv2d(x1, w.detach(), b, r, up=up), y1 + b + r, rtol=1e-5, atol=1e-5)
    # adjointness: <conv(x), g> == <x, dgrad(g)> == <w, wgrad(x, g)>
    xg = x1.clone().requires_grad_(True)
    g = rnd(y1.shape, 6)
    y = ops.conv2d(xg, w, None, up=up)
    y.backward(g)
    lhs = float((y.detach().double() * g.double()).sum())
    assert abs(float((xg.detach().double() * xg.grad.double()).sum()) - lhs) <= 1e-4 * abs(lhs) + 1e-3
    assert abs(float((w.detach().double() * w.grad.double()).sum()) - lhs) <= 1e-4 * abs(lhs) + 1e-3
    # determinism of the forward (no atomics): bitwise
    assert torch.equal(ops.conv2d(x1, w.detach(), zero_b, up=up), y1)


@pytest.mark.parametrize("H,C", [(32, 192), (16, 768), (8, 384)])
def test_group_norm_affine_invariance_full_size(ops, H, C):
    B = 128
    x = rnd((B, H, H, C), 7, 2.0)
    gam, bet = 1 + rnd((C,), 8, 0.2), rnd((C,), 9, 0.1)
    y = ops.group_norm_act(x, gam, bet, None, silu=False)
    y2 = ops.group_norm_act(3.0 * x + 5.0, gam, bet, None, silu=False)       # per-group shift/scale invariance
    torch.testing.assert_close(y2, y, rtol=1e-3, atol=1e-4)
    # a mean 1000x the spread: the shifted-sum / exact-merge moments (groupnorm.hip) must not lose the variance, which the
    # one-pass E[x^2] - mean^2 form does in fp32 (Welford / two-pass)
    xs = x[:8]
    y3 = ops.group_norm_act(xs + 2000.0, gam, bet, None, silu=False)
    torch.testing.assert_close(y3, y[:8], rtol=1e-3, atol=2e-3)      # the input itself carries only ~2e-4 absolute precision at 2000
    G = min(32, C // 4)
    z = ((y - bet) / gam).reshape(B, H * H, G, C // G)
    assert float(z.mean(dim=(1, 3)).abs().max()) < 1e-4
    assert float((z.var(dim=(1, 3), unbiased=False) - 1).abs().max()) < 1e-3


def test_attention_permutation_and_convexity_full_size(ops):
    B, h, heads = 128, 16, 6
    qkv = rnd((B, h, h, heads * 192), 10, 1.5)
    out = ops.attention(qkv, heads)
    perm = torch.randperm(h * h, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
    flat = qkv.reshape(B, h * h, heads, 3, 64)
    shuffled = flat.clone()
    shuffled[:, :, :, 1:] = flat[:, perm][:, :, :, 1:]        # permute keys and values together, keep queries
    out2 = ops.attention(shuffled.reshape(B, h, h, heads * 192), heads)
    torch.testing.assert_close(out2, out, rtol=1e-3, atol=1e-4)
    v = flat[:, :, :, 2]                                      # outputs are convex combinations of the values
    vmin, vmax = v.amin(dim=1, keepdim=True), v.amax(dim=1, keepdim=True)
    o = out.reshape(B, h * h, heads, 64)
    assert bool(((o >= vmin - 1e-4) & (o <= vmax + 1e-4)).all())


def test_full_model_eval_is_deterministic_and_batch_independent():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd.unet.uncond_unet import EDMPrecond
    torch.manual_seed(0)
    m = EDMPrecond(img_resolution=32, img_channels=3, model_channels=192, channel_mult=[1, 2, 2, 2], channel_mult_emb=4,
                   num_blocks=3, attn_resolutions=[16, 8], dropout=0.1, augment_dim=9).cuda().eval()
    with torch.no_grad():
        for n, p in m.named_parameters():          # wake up the zero-initialised branches
            if float(p.abs().max()) == 0:
                p.copy_(rnd(p.shape, hash(n) % 1000, 0.02))
        x, s = rnd((128, 3, 32, 32), 12), rnd((128,), 13, 0.49) + 0.5
        a = m(x, s)
        b = m(x, s)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])          # forward has no atomics: bitwise
        c = m(x[:16], s[:16])                                                # samples do not interact
        torch.testing.assert_close(c[0], a[0][:16], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(c[1], a[1][:16], rtol=1e-4, atol=1e-5)


def test_sampler_step_fixed_point_and_range(ops):
    n = 128 * 3 * 32 * 32
    x = rnd((n,), 14).double()
    zero = torch.zeros(n, device="cuda")
    y = ops.sampler_step(x.clone(), zero, zero, 0.6, 0.4, 0, True, 1.0, False)   # C = eps = 0 -> x is a fixed point
    assert torch.equal(y, x)
    z = ops.sampler_step((x * 3).clone(), zero, zero, 1e-4, 0.0, 0, True, 1.0, True)
    assert float(z.min()) >= 0.0 and float(z.max()) <= 1.0


@pytest.mark.parametrize("mode", ["x6", "2d", "1d"])
@pytest.mark.parametrize("B,H,cin,cout", [(128, 32, 192, 192), (128, 16, 768, 384), (128, 8, 384, 384)])
def test_winograd_and_direct_kernels_agree_full_size(ops, monkeypatch, B, H, cin, cout, mode):
    """At BASELINE's sizes the 3x3 convs run through the Winograd kernels -- by default 2-D F(2x2,3x3) with the f32 products on the
    bf16 MFMA by exact three-term splitting ("x6": conv_wino2d_x6.hip) and the f32-MFMA F(3x3,2x2) weight gradient, incl. the
    split-K small-map launches; ADM_BF16X6=0 -> the f32-MFMA 2-D kernels ("2d"); ADM_WINOGRAD2D=0 -> 1-D F(2,3) / F(3,2) ("1d").
    Forward, data gradient, weight gradient and bias gradient must agree with the direct implicit-GEMM kernels to fp32 rounding
    (different summation orders, same arithmetic)."""
    x = rnd((B, H, H, cin), 11)
    w0 = rnd((cout, cin, 3, 3), 12, 1 / math.sqrt(cin * 9))
    b0 = rnd((cout,), 13)
    gy = rnd((B, H, H, cout), 14)
    out = {}
    two_d = mode != "1d"
    monkeypatch.setattr(ops, "WINOGRAD2D", two_d)
    monkeypatch.setattr(ops, "BF16X6", mode == "x6")
    for wino in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD", wino)
        xd = x.clone().requires_grad_(True)
        w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        y = ops.conv2d(xd, w, b)
        pk = w._adm_packed
        assert (pk.w2f is not None) == (wino and two_d and mode != "x6") and (pk.wf is not None) == (wino and not two_d)
        assert (pk.w2f6 is not None) == (wino and mode == "x6")
        (y * gy).sum().backward()
        out[wino] = (y.detach(), xd.grad, w.grad, b.grad)
    for got, want, name in zip(out[True], out[False], ("y", "dx", "dw", "db")):
        err = float((got - want).abs().max() / want.abs().max())
        assert err <= (4e-5 if two_d else 2e-5), (name, err)


def test_full_model_training_step_winograd_vs_direct(ops, monkeypatch):
    """One loss + backward of the full 216 M-parameter model at bs=32 with the Winograd kernels on and off: the loss and
    the global gradient norm must agree to fp32 rounding, and the largest per-parameter gradient deviation stays small
    relative to that parameter's gradient norm."""
    from adm_amd.ddm.ddm_const import DDPM
    from adm_amd.unet.uncond_unet import EDMPrecond
    torch.manual_seed(0)
    unet = EDMPrecond(img_resolution=32, img_channels=3, model_channels=192, channel_mult=[1, 2, 2, 2], channel_mult_emb=4,
                      num_blocks=3, attn_resolutions=[16, 8], dropout=0.0, augment_dim=9)
    with torch.no_grad():
        for n, p in unet.named_parameters():
            if float(p.abs().max()) == 0:
                p.copy_(torch.empty_like(p).uniform_(-0.02, 0.02))
    dpm = DDPM(model=unet, image_size=[32, 32], sampling_timesteps=10, perceptual_weight=0.0,
               cfg=dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True)).cuda().train()
    x0, noise = rnd((32, 3, 32, 32), 21), rnd((32, 3, 32, 32), 22, 1.7)
    t = rnd((32,), 23, 0.45) + 0.5
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD", mode)
        dpm.zero_grad(set_to_none=True)
        loss, _ = dpm.training_step({"image": x0}, t=t, noise=noise)
        loss.backward()
        res[mode] = (float(loss.detach()), {n: p.grad.clone() for n, p in dpm.named_parameters() if p.grad is not None})
    (lw, gw), (ld, gd) = res[True], res[False]
    assert abs(lw - ld) <= 1e-5 * abs(ld)
    nw = math.sqrt(sum(float(g.double().pow(2).sum()) for g in gw.values()))
    nd = math.sqrt(sum(float(g.double().pow(2).sum()) for g in gd.values()))
    assert abs(nw - nd) <= 1e-4 * nd
    worst = max(float((gw[n] - gd[n]).norm() / (gd[n].norm() + 1e-6 * nd)) for n in gd)
    assert worst <= 2e-3, worst
