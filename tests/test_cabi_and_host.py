"""CPU-only: the C-ABI library loads and exports every symbol include/adm_hip.h declares; host logic
(module/state_dict layout, schedules, dotted-path aliases) matches the oracle and golden vectors.
No compute call is made here (no GPU)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import ddm_ref, unet_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(model_channels=64, num_blocks=1, dropout=0.0)


def test_library_exports_every_declared_symbol():
    from adm_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        hip.build()
    lib = hip.lib()
    header = open(os.path.join(ROOT, "include", "adm_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|long)\s+(adm_\w+)\s*\(", header, flags=re.M)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in adm_hip.h but not exported"
    assert sorted(hip.EXPORTS) == declared, set(hip.EXPORTS) ^ set(declared)
    assert lib.adm_version() >= 1
    assert lib.adm_gn_splits(1024, 192) == 16 and lib.adm_gn_splits(16, 384) == 1
    # the ctypes argument table has exactly one entry per parameter of the C declaration
    for m in re.finditer(r"^(?:int|long)\s+(adm_\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.M | re.S):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert len(hip._SIGS[name]) == n, f"{name}: header has {n} parameters, adm_amd/hip.py::_SIGS has {len(hip._SIGS[name])}"


def test_missing_library_fails_loudly(monkeypatch):
    from adm_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libadm_hip.so")
    with pytest.raises(RuntimeError, match="no non-HIP fallback"):
        hip.lib()


def test_ops_refuse_cpu_tensors():
    from adm_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.silu(torch.zeros(4))
    with pytest.raises(RuntimeError):
        ops.group_norm_act(torch.zeros(1, 2, 2, 32), torch.ones(32), torch.zeros(32))


@pytest.mark.parametrize("variant", unet_ref.VARIANTS)
def test_state_dict_layout_matches_reference_names(variant):
    """Same keys, shapes (incl. the 18 resample_filter buffers) as the reference's EDMPrecond: the oracle's
    param_shapes() was asserted equal to the imported reference's state_dict in tools/make_golden.py."""
    from ddm.utils import construct_class_by_name
    cfg = unet_ref.default_cfg(variant=variant, **SMALL)
    m = construct_class_by_name(class_name=f"unet.{variant}.EDMPrecond", img_resolution=32, img_channels=3,
                                model_type="DhariwalUNet", cfg={"ignored": 1},
                                **{k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks",
                                                       "attn_resolutions", "dropout", "augment_dim")})
    want = unet_ref.param_shapes(cfg)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(s) for k, s in want.items()}
    assert m.channels == 3 and m.self_condition is None and m.img_resolution == 32
    # reference init statistics: conv1 / proj / map_augment are zero, Dhariwal layers kaiming_uniform * sqrt(1/3)
    sd = m.state_dict()
    assert float(sd["model.enc.32x32_block0.conv1.weight"].abs().max()) == 0
    assert float(sd["model.map_augment.weight"].abs().max()) == 0
    w = sd["model.enc.32x32_block0.conv0.weight"]
    assert abs(float(w.std()) - (1 / (3 * w[0].numel())) ** 0.5) < 0.15 * (1 / (3 * w[0].numel())) ** 0.5


def test_full_width_parameter_count_on_meta_device():
    from adm_amd.unet.uncond_unet import EDMPrecond
    with torch.device("meta"):
        m = EDMPrecond(img_resolution=32, img_channels=3, model_channels=192, channel_mult=[1, 2, 2, 2],
                       channel_mult_emb=4, num_blocks=3, attn_resolutions=[16, 8], dropout=0.1, augment_dim=9)
    assert sum(p.numel() for p in m.parameters()) == 216141136
    assert len(m.state_dict()) == 829


def test_ddpm_wrapper_contract(golden_dir):
    from ddm.utils import construct_class_by_name
    from adm_amd.unet.uncond_unet import EDMPrecond
    cfg = unet_ref.default_cfg(**SMALL)
    unet = EDMPrecond(img_resolution=32, img_channels=3, **{k: cfg[k] for k in ("model_channels", "channel_mult", "num_blocks",
                                                                               "attn_resolutions", "dropout", "augment_dim")})
    model_cfg = dict(class_name="ddm.ddm_const.DDPM", image_size=[32, 32], ckpt_path=None, ignore_keys=[], only_model=False,
                     sampling_timesteps=10, loss_type="l2", start_dist="normal", perceptual_weight=0.0, eps=1e-4,
                     sigma_max=1, sigma_min=0.01, ldm=False, weighting_loss=True, use_l1=False, use_augment=True,
                     unet={"class_name": "unet.uncond_unet.EDMPrecond"})
    dpm = construct_class_by_name(model=unet, cfg=dict(model_cfg), **model_cfg)     # train_uncond_dpm.py:44-46 pattern
    assert dpm.image_size == [32, 32] and dpm.channels == 3 and dpm.sampling_timesteps == 10
    assert "eps" in dict(dpm.named_buffers()) and abs(float(dpm.eps) - 1e-4) < 1e-9
    assert list(dpm.state_dict().keys())[1].startswith("model.model.")
    g7 = np.load(os.path.join(golden_dir, "g7_samplers.npz"))
    np.testing.assert_allclose(dpm.t_steps().numpy(), g7["const.t_steps"], rtol=1e-12)
    t = torch.tensor([0.23, 0.81])
    w1, w2 = dpm.loss_weights(t)
    o1, o2 = ddm_ref.loss_weights("const", t, 1e-4)
    assert torch.equal(w1, o1) and torch.equal(w2, o2)
    import copy
    copy.deepcopy(dpm)                                       # EMA(model) deep-copies it (ddm/ema.py:77)
    with pytest.raises(AssertionError):
        construct_class_by_name(model=unet, class_name="ddm.ddm_const.DDPM", image_size=[32, 32], start_dist="cauchy")
    d2 = construct_class_by_name(model=unet, class_name="ddm.ddm_const_2.DDPM", image_size=[32, 32], perceptual_weight=0.0,
                                 cfg=dict(eps=1e-3, sigma_min=0.001, weighting_loss=True))
    np.testing.assert_allclose(d2.t_steps().numpy(), g7["const_2.t_steps"], rtol=1e-12)


def test_lr_and_ema_schedules(golden_dir):
    from adm_amd.optim import ema_decay_at, lr_lambda
    g9 = np.load(os.path.join(golden_dir, "g9_schedules.npz"))
    for s, d in zip(g9["ema_steps"], g9["ema_decay"]):
        assert abs(ema_decay_at(int(s)) - d) < 1e-12
    for i, r in zip(g9["lr_its"], g9["lr_ratio"]):
        assert abs(lr_lambda(int(i), 1e-4, 5e-6, 800000) - r) < 1e-12


def test_image_stream_never_trains_on_noise_silently(tmp_path):
    """a data section that names a real dataset must load it or raise; U(-1,1) noise only for 'synthetic'."""
    import pickle
    from train_uncond_dpm import ImageStream
    dev = torch.device("cpu")
    s = ImageStream({"class_name": "synthetic"}, 4, (32, 32), dev, 0)
    assert next(s)["image"].shape == (4, 3, 32, 32)
    with pytest.raises(NotImplementedError):
        ImageStream({"class_name": "ddm.data.CelebAHQ", "img_folder": "/x"}, 4, (32, 32), dev, 0)
    with pytest.raises(NotImplementedError):
        ImageStream({}, 4, (32, 32), dev, 0)
    with pytest.raises(FileNotFoundError):
        ImageStream({"class_name": "synthetic", "npy": str(tmp_path / "missing.npy")}, 4, (32, 32), dev, 0)
    np.save(tmp_path / "bad.npy", np.zeros((5, 16, 16, 3), np.uint8))
    with pytest.raises(ValueError):
        ImageStream({"npy": str(tmp_path / "bad.npy")}, 4, (32, 32), dev, 0)
    np.save(tmp_path / "f32.npy", np.zeros((5, 32, 32, 3), np.float32))
    with pytest.raises(ValueError):
        ImageStream({"npy": str(tmp_path / "f32.npy")}, 4, (32, 32), dev, 0)
    arr = (np.arange(6 * 32 * 32 * 3) % 251).astype(np.uint8).reshape(6, 32, 32, 3)
    np.save(tmp_path / "ok.npy", arr)
    s = ImageStream({"npy": str(tmp_path / "ok.npy"), "augment_horizontal_flip": False}, 4, (32, 32), dev, 0)
    b = next(s)["image"]
    assert b.shape == (4, 3, 32, 32) and float(b.min()) >= -1 and float(b.max()) <= 1
    # the reference's YAML: class_name ddm.data.CIFAR10 + img_folder with the standard python batches
    base = tmp_path / "cifar-10-batches-py"
    base.mkdir()
    for i in range(1, 6):
        data = ((np.arange(3 * 3072) + i) % 256).astype(np.uint8).reshape(3, 3072)
        with open(base / f"data_batch_{i}", "wb") as f:
            pickle.dump({"data": data, "labels": [0, 1, 2]}, f)
    s = ImageStream({"class_name": "ddm.data.CIFAR10", "img_folder": str(tmp_path)}, 4, (32, 32), dev, 0)
    assert s.images.shape == (15, 3, 32, 32)
    first = ((np.arange(3072) + 1) % 256).astype(np.float32).reshape(3, 32, 32) / 127.5 - 1      # CHW planes, data.py:96-97
    np.testing.assert_allclose(s.images[0].numpy(), first, atol=1e-6)
    with pytest.raises(FileNotFoundError):
        ImageStream({"class_name": "ddm.data.CIFAR10", "img_folder": str(tmp_path / "nowhere")}, 4, (32, 32), dev, 0)


def test_warmup_iter_is_configurable():
    from adm_amd.optim import lr_lambda
    assert lr_lambda(99, 1e-4, 5e-6, 1000, warmup=100) == 1.0
    assert lr_lambda(49, 1e-4, 5e-6, 1000, warmup=100) == 0.5
    assert abs(lr_lambda(600, 1e-4, 5e-6, 1000, warmup=100) - 0.5 ** 0.96) < 1e-12


def test_deferred_gradient_tables_host_logic(monkeypatch):
    """Host side of the end-of-backward gradient tables (ops._defer_unpack / _defer_gn_param / flush_deferred_unpack), no GPU:
    the rows and their block prefix sums, one flush per backward pass, the table cache, and the clean-up after a pass that raised."""
    from adm_amd import ops
    calls = []
    monkeypatch.setattr(ops, "call", lambda name, *a: calls.append((name, a)))
    monkeypatch.setattr(ops, "ptr", lambda t: t)
    ops.reset_deferred_unpack()
    monkeypatch.setattr(ops, "_rest_ws", {})
    monkeypatch.setattr(ops, "_unpack_tables", {})
    monkeypatch.setattr(ops, "_gn_tables", {})
    ws1, ws2 = torch.ones(64, 9 * 32), torch.ones(32, 12 * 64)
    ops._rest_ws[("a", 0)], ops._rest_ws[("b", 0)] = ws1, ws2
    g1, g2 = torch.zeros(40, 24, 3, 3), torch.zeros(32, 47, 3, 3)
    red, dgam, dbet = torch.zeros(1000), torch.zeros(96), torch.zeros(96)

    class Layer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fail):
            ctx.fail = fail
            return x * 2

        @staticmethod
        def backward(ctx, g):
            ops._begin_defer()
            ops._defer_unpack(ws1, g1, 40, 24, 9, 32, 0)            # plain 3x3 layout: 40*24*9 = 8640 items -> 5 blocks
            ops._defer_gn_param(red, 200, None, 0, dgam, dbet, 4, 96)   # 96 channels -> 3 blocks
            if ctx.fail:
                raise RuntimeError("boom")
            ops._defer_unpack(ws2, g2, 32, 47, 0, 64, 0)            # Winograd planes: 32*47*3 = 4512 items -> 3 blocks
            return g * 2, None

    x = torch.ones(3, requires_grad=True)
    Layer.apply(x, False).sum().backward()
    names = [c[0] for c in calls]
    assert names == ["adm_gn_bwd_param_table", "adm_unpack_wgrad_table"]
    gn_table, gn_rows, gn_blocks = calls[0][1]
    assert gn_rows == 1 and gn_blocks == 3
    assert gn_table.tolist() == [[red.data_ptr() + 800, 0, 0, dgam.data_ptr(), dbet.data_ptr(), 4, 96, 0]]
    table, rows, blocks = calls[1][1]
    assert rows == 2 and blocks == 5 + 3
    t = table.tolist()
    assert t[0][:10] == [ws1.data_ptr(), g1.data_ptr(), 40, 24, 9, 32, 0, 1, 1, 0]
    assert t[1][:10] == [ws2.data_ptr(), g2.data_ptr(), 32, 47, 0, 64, 0, 1, 1, 5]
    assert not ops._unpack_rows and not ops._gn_rows and not ops._unpack_keep
    # a second pass with the same rows: one flush again, the cached tables, no new upload
    uploads = ops.table_uploads
    del calls[:]
    Layer.apply(x, False).sum().backward()
    assert [c[0] for c in calls] == ["adm_gn_bwd_param_table", "adm_unpack_wgrad_table"] and ops.table_uploads == uploads
    assert calls[1][1][0] is table
    # a pass that raises leaves its rows behind; the next pass drops them and re-zeroes every workspace BEFORE it queues its own
    del calls[:]
    with pytest.raises(RuntimeError):
        Layer.apply(x, True).sum().backward()
    assert len(ops._unpack_rows) == 1 and len(ops._gn_rows) == 1 and not calls
    assert float(ws1.abs().max()) == 1.0
    Layer.apply(x, False).sum().backward()
    assert float(ws1.abs().max()) == 0.0 and float(ws2.abs().max()) == 0.0
    assert [c[0] for c in calls] == ["adm_gn_bwd_param_table", "adm_unpack_wgrad_table"]
    assert calls[1][1][1] == 2 and not ops._unpack_rows and not ops._gn_rows
    ops.reset_deferred_unpack()


def test_selection_context_and_use_counts_host_logic():
    """CPU: (1) ops.batch_invariant(): kernel selection by the WHOLE batch's pixel count, whatever the size of the pass (the frozen
    autoencoder's chunks must get the bits of the unchunked call); (2) ops._mark_uses / ops._notify: a parameter that takes part in
    the graph twice is announced to its gradient sink once, by its LAST backward node, and FlatParams.zero_grad resets the count."""
    from adm_amd import ops

    class Ctx:
        needs_input_grad = (True, True, True)

    assert ops._use_wino(1, 32, 32, 3, False, -1) == (1024 >= ops.WINO_MIN_M)
    with ops.batch_invariant(8):
        assert ops._use_wino(1, 32, 32, 3, False, -1) == ops._use_wino(8, 32, 32, 3, False, -1) == (8192 >= ops.WINO_MIN_M)
        assert ops._sel_batch(2) == 8 and ops._sel_batch(16) == 16
        with ops.batch_invariant(32):
            assert ops._sel_batch(2) == 32
        assert ops._sel_batch(2) == 8
    assert ops._sel_batch(2) == 2
    p = torch.nn.Parameter(torch.zeros(4))
    p._adm_direct = True
    heard = []
    p._adm_grad_sink = lambda q: heard.append(q)
    ops._mark_uses(Ctx, (1, p)); ops._mark_uses(Ctx, (1, p)); ops._mark_uses(Ctx, (2, None))
    assert p._adm_uses == 2
    ops._notify(p)
    assert heard == [] and p._adm_uses == 1
    ops._notify(p)
    assert heard == [p] and p._adm_uses == 0
    ops._notify(p)                                   # (a use that was never counted, e.g. recorded before FlatParams existed)
    assert heard == [p, p]
    q = torch.nn.Parameter(torch.zeros(4))           # not a direct-gradient parameter: never counted
    ops._mark_uses(Ctx, (1, q))
    assert not hasattr(q, "_adm_uses")
