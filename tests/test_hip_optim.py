"""GPU: flat-buffer training state.  (1) gradients accumulated directly into the flat buffer by the HIP
backward kernels equal the autograd-returned gradients; (2) the fused clip + AdamW + EMA kernel equals
torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW + lerp (train_uncond_dpm.py:292-310, ddm/ema.py:186)."""
import copy

import pytest
import torch

from oracle import fill, unet_ref

pytestmark = pytest.mark.gpu
SMALL = dict(model_channels=64, num_blocks=1, dropout=0.0)


def _model(gpu):
    from adm_amd.unet.uncond_unet import EDMPrecond
    cfg = unet_ref.default_cfg(variant="uncond_unet", **SMALL)
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions",
                              "dropout", "augment_dim")}
    m = EDMPrecond(img_resolution=32, img_channels=3, **kw)
    m.load_state_dict(fill.filled_state_dict(unet_ref.param_shapes(cfg)))
    return m.to(gpu).eval()


def _loss(m, gpu):
    x = fill.hash_tensor((2, 3, 32, 32), "x", 1.0).to(gpu)
    sigma = torch.tensor([0.05, 0.7], device=gpu)
    dx, dy = m(x, sigma)
    return (dx * fill.hash_tensor(dx.shape, "gx", 1.0).to(gpu)).sum() + (dy * fill.hash_tensor(dy.shape, "gy", 1.0).to(gpu)).sum()


def test_direct_flat_gradients_and_fused_adamw():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd.optim import FlatParams, FusedAdamWEMA
    gpu = torch.device("cuda:0")
    ref = _model(gpu)
    _loss(ref, gpu).backward()                     # plain autograd path (no flat buffers)
    m = _model(gpu)
    flat = FlatParams(m)
    for _ in range(2):                             # second pass: zero_grad must fully reset the direct sinks
        flat.zero_grad()
        _loss(m, gpu).backward()
    names = [n for n, _ in m.named_parameters()]
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offsets[names.index(n)]
        if q.grad is None:          # parameter unused by this loss (map_augment without labels)
            assert float(p.grad.abs().max()) == 0, n
            continue
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-4, atol=1e-5 * float(q.grad.abs().max()) + 1e-12, msg=n)

    # reference optimiser on CPU copies
    # (the CPU optimiser is fed the flat buffer's gradients bit for bit: Adam's m/sqrt(v) turns a sign flip of
    #  a ~0 gradient -- e.g. from split-K atomics ordering -- into a full +-lr step, which is not what is tested)
    cpu = copy.deepcopy(ref).cpu()
    g_cpu = [p.grad.detach().cpu().clone() for p in m.parameters()]
    for pc, g in zip(cpu.parameters(), g_cpu):
        pc.grad = g.clone()
    ema_ref = [p.detach().clone() for p in cpu.parameters()]
    topt = torch.optim.AdamW(cpu.parameters(), lr=3e-4, weight_decay=1e-2)
    opt = FusedAdamWEMA(flat, lr=3e-4, weight_decay=1e-2, max_norm=1.0, ema=True)
    for step in range(3):
        total = torch.nn.utils.clip_grad_norm_(cpu.parameters(), 1.0)
        topt.step()
        for e, p in zip(ema_ref, cpu.parameters()):
            e.lerp_(p.detach(), 1 - 0.9)
        opt.step(ema_decay=0.9)
        assert abs(opt.grad_norm() - float(total)) <= 1e-4 * float(total)
        # both sides keep the SAME gradients for the next step (clip_grad_norm_ scaled the CPU copy in place)
        for pc, g in zip(cpu.parameters(), g_cpu):
            pc.grad = g.clone()
    for (n, p), pc, e in zip(m.named_parameters(), cpu.parameters(), ema_ref):
        torch.testing.assert_close(p.detach().cpu(), pc.detach(), rtol=1e-4, atol=1e-6, msg=n)
        o = flat.offsets[names.index(n)]
        torch.testing.assert_close(opt.ema[o:o + p.numel()].view(p.shape).cpu(), e, rtol=1e-4, atol=1e-6, msg=n)
    # packed weights must have been invalidated: a forward now uses the updated parameters
    y_new = m(fill.hash_tensor((2, 3, 32, 32), "x", 1.0).to(gpu), torch.tensor([0.05, 0.7], device=gpu))[0]
    m2 = _model(gpu)
    m2.load_state_dict(m.state_dict())
    y_chk = m2(fill.hash_tensor((2, 3, 32, 32), "x", 1.0).to(gpu), torch.tensor([0.05, 0.7], device=gpu))[0]
    torch.testing.assert_close(y_new, y_chk, rtol=1e-5, atol=1e-6)


def test_training_reduces_the_loss_on_a_fixed_batch():
    """End-to-end sanity of forward + backward + fused optimiser: overfit one batch with a reduced UNet (dropout on,
    reference init incl. the zero-initialised conv1 / proj / map_augment)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd.ddm.ddm_const import DDPM
    from adm_amd.optim import FlatParams, FusedAdamWEMA
    from adm_amd.unet.uncond_unet import EDMPrecond
    torch.manual_seed(0)
    gpu = torch.device("cuda:0")
    unet = EDMPrecond(img_resolution=32, img_channels=3, model_channels=64, channel_mult=[1, 2, 2, 2], num_blocks=1,
                      attn_resolutions=[16, 8], dropout=0.1, augment_dim=9)
    dpm = DDPM(model=unet, image_size=[32, 32], perceptual_weight=0.0, cfg=dict(eps=1e-4, weighting_loss=False)).to(gpu).train()
    flat = FlatParams(dpm)
    opt = FusedAdamWEMA(flat, lr=2e-3, weight_decay=0.0, max_norm=1.0, ema=True)
    g = torch.Generator(device=gpu).manual_seed(1)
    batch = {"image": torch.rand(16, 3, 32, 32, device=gpu, generator=g) * 2 - 1}
    t = torch.rand(16, device=gpu, generator=g) * 0.8 + 0.1
    noise = torch.randn(16, 3, 32, 32, device=gpu, generator=g)
    losses = []
    for step in range(40):
        flat.zero_grad()
        loss, _ = dpm.training_step(batch, t=t, noise=noise)
        loss.backward()
        opt.step(ema_decay=0.9)
        losses.append(float(loss.detach()))
    assert all(l == l for l in losses), "NaN in the loss"
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    # the EMA trails the weights but has moved away from the initial copy
    assert float((opt.ema - flat.flat).abs().max()) > 0


def test_table_repack_equals_per_layer_pack():
    """repack_all() (one tiled launch over every layer, run after each optimiser step) must reproduce adm_pack_weight's
    operands bit for bit for every registered parameter: 3x3 / 1x1 convs, Linears, the 3-channel stem (Ci padded to 32),
    the 3-channel heads (Co padded) and the qkv convs (row permutation)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import ops
    from adm_amd.hip import call, ptr
    gpu = torch.device("cuda:0")
    m = _model(gpu)
    old_min, ops.WINO_MIN_M = ops.WINO_MIN_M, 1    # batch 2: force the 3x3 layers through Winograd so their operands exist
    old_x6, ops.BF16X6 = ops.BF16X6, False         # f32-MFMA kernels: 1-D operands for the up-sampling layers, 2-D for the others
    try:                                           # (the split-bf16 images: tests/test_hip_ops.py::test_conv_x6_weights_follow_repack_all)
        _loss(m, gpu).backward()                   # populates the packed-operand cache for every layer
    finally:
        ops.WINO_MIN_M, ops.BF16X6 = old_min, old_x6
    ents = [(w(), ks, qkv) for w, _b, ks, qkv in ops._pack_registry.values() if w() is not None and w().is_cuda]
    mine = [(w, ks, qkv) for (w, ks, qkv) in ents if any(w is p for p in m.parameters())]
    assert len(mine) > 50
    with torch.no_grad():
        for w, _, _ in mine:                       # rewrite the weights behind the cache's back (as the fused optimiser does)
            w.view(-1).copy_(fill.hash_tensor((w.numel(),), "repack", 0.3).to(gpu))
    ops.repack_all()
    kinds = set()
    for w, ks, qkv in mine:
        ent = w._adm_packed
        co, ci = w.shape[0], w.shape[1]
        cop, cip = ops.ceil32(co), ops.ceil32(ci)
        f, b = torch.empty_like(ent.fwd), torch.empty_like(ent.bwd)
        call("adm_pack_weight", ptr(w.detach()), ptr(f), ptr(b), co, ci, ks, cop, cip, int(qkv))
        assert torch.equal(f, ent.fwd) and torch.equal(b, ent.bwd), (tuple(w.shape), ks, qkv)
        if ent.wf is not None:                     # Winograd operands of the 3x3 layers ride in the same launch
            wf, wb = torch.empty_like(ent.wf), torch.empty_like(ent.wb)
            call("adm_pack_weight_wino", ptr(w.detach()), ptr(wf), ptr(wb), co, ci, cop, cip)
            assert torch.equal(wf, ent.wf) and torch.equal(wb, ent.wb), tuple(w.shape)
            kinds.add("wino")
        if ent.w2f is not None:                    # ... and the 2-D Winograd planes
            w2f, w2b = torch.empty_like(ent.w2f), torch.empty_like(ent.w2b)
            call("adm_pack_weight_wino2d", ptr(w.detach()), ptr(w2f), ptr(w2b), co, ci, cop, cip)
            assert torch.equal(w2f, ent.w2f) and torch.equal(w2b, ent.w2b), tuple(w.shape)
            kinds.add("wino2d")
        kinds.add((ks, bool(qkv), cop != co, cip != ci))
    assert {(3, False, False, True), (3, False, True, False), (1, True, False, False), (1, False, False, False), "wino", "wino2d"} <= kinds
