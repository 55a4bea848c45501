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


def test_unpack_table_equals_per_layer_unpack_and_clears_the_workspaces():
    """adm_unpack_wgrad_table (one launch for all layers, zero-at-rest workspaces) against adm_unpack_wgrad /
    adm_unpack_wgrad_wino2d on the same packed tiles: bit-identical gradients, every element it read is cleared."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import hip
    gpu = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    layers = [(64, 96, 9, 0), (192, 64, 1, 1), (40, 24, 1, 0), (96, 64, 0, 0), (33, 47, 9, 0), (64, 64, 0, 0)]   # (co, ci, taps | 0 = wino2d planes, qkv)
    rows, keep, want, begin = [], [], [], 0
    for co, ci, taps, qkv in layers:
        cop, cip = (co + 31) // 32 * 32, (ci + 31) // 32 * 32
        ws = torch.randn((cop, (taps if taps else 12) * cip), generator=g).to(gpu)
        ks = 3 if taps in (0, 9) else 1
        dst = torch.randn((co, ci, ks, ks), generator=g).to(gpu)
        ref = dst.clone()
        if taps:
            hip.call("adm_unpack_wgrad", hip.ptr(ws), hip.ptr(ref), co, ci, ks, cop, cip, qkv, 1)
        else:
            hip.call("adm_unpack_wgrad_wino2d", hip.ptr(ws), 1, hip.ptr(ref), co, ci, cop, cip, 1, None, None)
        items = co * ci * (taps if taps else 3)
        rows.append([ws.data_ptr(), dst.data_ptr(), co, ci, taps, cip, qkv, 1, 1, begin, 0, 0])
        begin += (items + 2047) // 2048
        keep.append((ws, dst, cop, cip, co, ci, taps))
        want.append(ref)
    table = torch.tensor(rows, dtype=torch.int64, device=gpu)
    hip.call("adm_unpack_wgrad_table", hip.ptr(table), len(rows), begin)
    torch.cuda.synchronize()
    for (ws, dst, cop, cip, co, ci, taps), ref in zip(keep, want):
        assert torch.equal(dst, ref)
        planes = ws.view(cop, taps if taps else 12, cip)
        assert float(planes[:co, :, :ci].abs().max()) == 0.0      # (the qkv row's packed rows are a permutation of 0..co-1)


def test_deferred_unpack_matches_per_layer_launches():
    """A backward pass with the end-of-backward table unpack (default) against ADM_DEFER_UNPACK=0: same gradients (split-K
    atomics reorder sums: 1e-5 relative), and every zero-at-rest workspace is zero again afterwards."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import ops
    from adm_amd.optim import FlatParams
    gpu = torch.device("cuda:0")
    grads = []
    old = ops.DEFER_UNPACK
    try:
        for defer in (False, True):
            ops.DEFER_UNPACK = defer
            m = _model(gpu)
            flat = FlatParams(m)
            for _ in range(2):
                flat.zero_grad()
                _loss(m, gpu).backward()
            torch.cuda.synchronize()
            grads.append(flat.grad.clone())
            if defer:
                assert ops._rest_ws and not ops._unpack_rows
                for ws in ops._rest_ws.values():
                    assert float(ws.abs().max()) == 0.0
    finally:
        ops.DEFER_UNPACK = old
    scale = float(grads[0].abs().max())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale


def test_side_stream_weight_gradients_match_single_stream():
    """ops.SIDE_WGRAD (weight / bias gradients on a side stream, joined before the end-of-backward unpack) against the
    single-stream order: same gradients over repeated steps (split-K atomics reorder sums: 1e-5 relative), workspaces zero."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import ops
    from adm_amd.optim import FlatParams
    gpu = torch.device("cuda:0")
    grads = []
    old = ops.SIDE_WGRAD
    try:
        for side in (False, True):
            ops.SIDE_WGRAD = side
            m = _model(gpu)
            flat = FlatParams(m)
            for _ in range(3):
                flat.zero_grad()
                _loss(m, gpu).backward()
            torch.cuda.synchronize()
            grads.append(flat.grad.clone())
            for ws in ops._rest_ws.values():
                assert float(ws.abs().max()) == 0.0
    finally:
        ops.SIDE_WGRAD = old
    scale = float(grads[0].abs().max())
    assert scale > 0 and bool(torch.isfinite(grads[1]).all())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale


def test_second_decoder_on_its_own_stream_matches_single_stream():
    """ops.BRANCH_STREAM (the second decoder of the two-output UNet on its own stream, forward and backward) against one stream:
    bit-identical outputs, gradients equal up to the split-K atomics' summation order, over repeated steps."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import ops
    from adm_amd.optim import FlatParams
    gpu = torch.device("cuda:0")
    x = fill.hash_tensor((2, 3, 32, 32), "x", 1.0).to(gpu)
    sigma = torch.tensor([0.05, 0.7], device=gpu)
    outs, grads = [], []
    old = ops.BRANCH_STREAM
    try:
        for branch in (False, True):
            ops.BRANCH_STREAM = branch
            m = _model(gpu)
            with torch.no_grad():
                for _ in range(2):
                    dx, dy = m(x, sigma)
            torch.cuda.synchronize()
            outs.append((dx.clone(), dy.clone()))
            flat = FlatParams(m)
            for _ in range(3):
                flat.zero_grad()
                _loss(m, gpu).backward()
            torch.cuda.synchronize()
            grads.append(flat.grad.clone())
    finally:
        ops.BRANCH_STREAM = old
    assert ops._branch_stream is not None
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    scale = float(grads[0].abs().max())
    assert scale > 0 and bool(torch.isfinite(grads[1]).all())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale


def test_backward_that_raises_leaves_no_stale_gradient_rows():
    """A backward pass that raises half-way leaves queued unpack rows and dirty workspaces behind; the next pass must drop them
    (not add them to the gradients, and not stay un-flushed): its gradients equal those of an undisturbed model."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import ops
    from adm_amd.optim import FlatParams
    gpu = torch.device("cuda:0")

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return a.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    ref = _model(gpu)
    fr = FlatParams(ref)
    fr.zero_grad()
    _loss(ref, gpu).backward()
    torch.cuda.synchronize()

    m = _model(gpu)
    flat = FlatParams(m)
    flat.zero_grad()
    # the first encoder layer's output passes through the failing node: everything behind it runs its backward, then the pass dies
    first = next(iter(m.model.enc.values()))
    orig = first.forward
    first.forward = lambda x, *a, **k: Boom.apply(orig(x, *a, **k))
    x = fill.hash_tensor((2, 3, 32, 32), "x", 1.0).to(gpu)
    try:
        dx, dy = m(x, torch.tensor([0.05, 0.7], device=gpu))
        with pytest.raises(RuntimeError):
            ((dx * 1.5).sum() + (dy * 0.5).sum()).backward()
    finally:
        first.forward = orig
    torch.cuda.synchronize()
    assert ops._unpack_rows or ops._gn_rows            # the dead pass left rows behind
    flat.zero_grad()
    _loss(m, gpu).backward()
    torch.cuda.synchronize()
    assert not ops._unpack_rows and not ops._gn_rows
    scale = float(fr.grad.abs().max())
    assert float((flat.grad - fr.grad).abs().max()) <= 1e-5 * scale


def test_module_applied_twice_with_direct_gradients_and_a_reducer_sink():
    """A conv and a GroupNorm applied TWICE in one forward pass, gradients accumulated directly into the flat buffer through the
    deferred tables: the second weight-gradient kernel must not meet its own pending tile (plain stores with one split),
    the two GroupNorm rows must not race on dgamma / dbeta, and a gradient sink (the bucketed reducer's hook) hears about each
    parameter ONCE, after its last backward node -- autograd's accumulate-grad semantics."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.nn as nn
    from adm_amd import ops
    from adm_amd.optim import FlatParams
    gpu = torch.device("cuda:0")

    class Twice(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Parameter(fill.hash_tensor((64, 64, 3, 3), "tw.w", 0.05))
            self.b = nn.Parameter(fill.hash_tensor((64,), "tw.b", 0.1))
            self.g = nn.Parameter(1 + fill.hash_tensor((64,), "tw.g", 0.1))
            self.be = nn.Parameter(fill.hash_tensor((64,), "tw.be", 0.1))

        def forward(self, x):
            for _ in range(2):
                x = ops.conv2d(ops.group_norm_act(x, self.g, self.be, None, silu=True), self.w, self.b)
            return x

    x = fill.hash_tensor((4, 16, 16, 64), "tw.x", 1.0).to(gpu)
    gy = fill.hash_tensor((4, 16, 16, 64), "tw.gy", 1.0).to(gpu)
    ref = Twice().to(gpu)
    (ref(x) * gy).sum().backward()                 # plain autograd: every use returns its own gradient, autograd adds them
    m = Twice().to(gpu)
    flat = FlatParams(m)
    heard = []
    for p in m.parameters():
        p._adm_grad_sink = (lambda q, _p=p: heard.append(id(_p)))
    for _ in range(2):
        flat.zero_grad()
        heard.clear()
        (m(x) * gy).sum().backward()
    torch.cuda.synchronize()
    assert sorted(heard) == sorted(id(p) for p in m.parameters()), "each parameter is announced once, after its last use"
    for (n, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-4, atol=1e-5 * float(q.grad.abs().max()), msg=n)


def test_fanout_sums_the_gradients_of_all_consumers():
    """ops.fanout: n aliases of a tensor; backward = one add3 launch per three gradients (odd sizes: the two-input kernel)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import ops
    gpu = torch.device("cuda:0")
    for shape, n in (((2, 8, 8, 32), 3), ((2, 8, 8, 32), 2), ((3, 5, 7), 4), ((2, 4, 4, 64), 5)):
        x = fill.hash_tensor(shape, "fan.x", 1.0).to(gpu).requires_grad_(True)
        ws = [fill.hash_tensor(shape, f"fan.w{i}", 1.0).to(gpu) for i in range(n)]
        outs = ops.fanout(x, n)
        assert len(outs) == n and all(o.data_ptr() == x.data_ptr() for o in outs)
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        torch.testing.assert_close(x.grad, sum(ws), rtol=1e-6, atol=1e-6)
    with torch.no_grad():
        assert all(o is x for o in ops.fanout(x, 3))
    y = fill.hash_tensor((2, 8, 8, 32), "fan.y", 1.0).to(gpu).requires_grad_(True)
    a, b, c = ops.fanout(y, 3)                     # a consumer that never contributes a gradient
    ((a * 2).sum() + (c * 3).sum()).backward()
    torch.testing.assert_close(y.grad, torch.full_like(y, 5.0))
