"""GPU parity tests, model level: the HIP UNet / DDPM (through the reference-shaped Python API) against
(a) the committed golden vectors the imported reference produced and (b) the CPU oracle on the same
inputs.  rtol 1e-3 / atol 1e-4 fp32 (north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import ddm_ref, fill, unet_ref

from parity import close  # noqa: E402  (tests/parity.py: the north_star tolerance, elementwise)

pytestmark = pytest.mark.gpu
SMALL = dict(model_channels=64, num_blocks=1, dropout=0.0)
RTOL, ATOL = 1e-3, 1e-4


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def build_unet(variant, gpu, full=False, **over):
    import importlib
    mod = importlib.import_module("unet." + variant)          # the reference's dotted path (alias package)
    cfg = unet_ref.default_cfg(variant=variant, **({} if full else SMALL))
    cfg.update(over)
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions",
                              "dropout", "augment_dim")}
    m = mod.EDMPrecond(img_resolution=32, img_channels=3, model_type="DhariwalUNet", **kw)
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    m.load_state_dict(sd, strict=True)
    return m.to(gpu), cfg, sd


def small_inputs(cfg, B=2):
    x = fill.hash_tensor((B, 3, 32, 32), "x", 1.0)
    sigma = torch.tensor([0.05, 0.7, 0.31, 0.999][:B])
    aug = fill.hash_tensor((B, cfg["augment_dim"]), "xaug", 1.0)
    return x, sigma, aug


@pytest.mark.parametrize("variant", unet_ref.VARIANTS)
@pytest.mark.parametrize("use_aug", [0, 1])
def test_precond_small_vs_golden(gpu, golden_dir, variant, use_aug):
    g = np.load(os.path.join(golden_dir, "g5_precond_small.npz"))
    m, cfg, _ = build_unet(variant, gpu)
    m.eval()
    x, sigma, aug = small_inputs(cfg)
    kw = dict(augment_labels=aug.to(gpu)) if use_aug else {}
    xg = x.to(gpu).requires_grad_(True)            # the denoiser is differentiable in its input too (uncond_unet.py:614-635)
    dx, dy = m(xg, sigma.to(gpu), **kw)
    p = f"{variant}.aug{use_aug}."
    assert dx.dtype == torch.float32 and dx.shape == (2, 3, 32, 32)
    close(dx, g[p + "D_x"]); close(dy, g[p + "D_y"])
    gx, gy = fill.hash_tensor(dx.shape, "gx", 1.0).to(gpu), fill.hash_tensor(dy.shape, "gy", 1.0).to(gpu)
    ((dx * gx).sum() + (dy * gy).sum()).backward()
    assert xg.grad is not None and xg.grad.shape == x.shape and xg.grad.dtype == torch.float32
    close(xg.grad, g[p + "dL_dx"])                 # the reference's own dL/dx_t on the same inputs
    named = dict(m.named_parameters())
    for key in g.files:
        if key.startswith(p + "grad."):
            name = key[len(p + "grad."):]
            got = named[name].grad.reshape(-1)[:4096]
            close(got, g[key], scale=float(g[p + "gradnorm." + name]) / max(1.0, np.sqrt(min(got.numel(), 4096)) / 8))
            gn = float(named[name].grad.double().norm())
            assert abs(gn - float(g[p + "gradnorm." + name])) <= 2e-3 * float(g[p + "gradnorm." + name]) + 1e-6, name


@pytest.mark.parametrize("winograd", [False, True])
def test_every_parameter_gets_the_oracle_gradient(gpu, monkeypatch, winograd):
    """All 400+ parameter gradients of the reduced two-decoder model vs autograd through the CPU oracle -- once with every
    3x3 conv on the direct implicit GEMM and once with every eligible one forced through the Winograd F(2,3) kernel (at
    this test's batch of 2 the size threshold would otherwise keep them all on the direct kernel)."""
    from adm_amd import ops
    monkeypatch.setattr(ops, "WINOGRAD", winograd)
    monkeypatch.setattr(ops, "WINO_MIN_M", 1)
    m, cfg, sd = build_unet("uncond_unet", gpu)
    m.eval()
    x, sigma, aug = small_inputs(cfg)
    dx, dy = m(x.to(gpu), sigma.to(gpu), augment_labels=aug.to(gpu))
    gx, gy = fill.hash_tensor(dx.shape, "gx", 1.0), fill.hash_tensor(dy.shape, "gy", 1.0)
    ((dx * gx.to(gpu)).sum() + (dy * gy.to(gpu)).sum()).backward()
    sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd.items()}
    ox, oy = unet_ref.edm_precond(sdo, cfg, x, sigma, augment_labels=aug)
    close(dx, ox.detach()); close(dy, oy.detach())
    ((ox * gx).sum() + (oy * gy).sum()).backward()
    bad = []
    gmax = max(float(v.grad.double().norm()) for v in sdo.values() if v.grad is not None)
    for name, p in m.named_parameters():
        want = sdo[name].grad
        assert p.grad is not None, name
        # floor: k_conv.bias has an analytically ZERO gradient (softmax shift invariance), so only noise is left
        err = float((p.grad.cpu().double() - want.double()).norm() / (want.double().norm() + 1e-6 * gmax))
        if err > 2e-3:
            bad.append((name, err))
    assert not bad, bad[:10]


def test_scalar_fp64_sigma_and_fp64_input(gpu, golden_dir):
    g = np.load(os.path.join(golden_dir, "g5b_scalar_sigma.npz"))
    m, cfg, _ = build_unet("uncond_unet", gpu)
    m.eval()
    x, _, _ = small_inputs(cfg)
    with torch.no_grad():
        dx, dy = m(x.double().to(gpu), torch.tensor(0.37, dtype=torch.float64, device=gpu))
    assert dx.dtype == torch.float32
    close(dx, g["D_x"]); close(dy, g["D_y"])


def test_full_width_cifar_model_vs_golden(gpu, golden_dir):
    """The 216 M-parameter CIFAR-10 configuration (BASELINE configs[1] network), B=1."""
    g = np.load(os.path.join(golden_dir, "g5c_full_width.npz"))
    m, cfg, _ = build_unet("uncond_unet", gpu, full=True, dropout=0.0)
    assert sum(p.numel() for p in m.parameters()) == 216141136
    m.eval()
    x = fill.hash_tensor((1, 3, 32, 32), "x", 1.0)
    sigma = torch.tensor([0.05])
    aug = fill.hash_tensor((1, 9), "xaug", 1.0)
    with torch.no_grad():
        dx, dy = m(x.to(gpu), sigma.to(gpu), augment_labels=aug.to(gpu))
    close(dx, g["D_x"]); close(dy, g["D_y"])


G2_CLASSES = [  # cin, cout, Hin, up, down, attn  (tools/make_golden.py BLOCK_CLASSES)
    (192, 192, 32, 0, 0, 0), (384, 192, 32, 0, 0, 0), (576, 192, 32, 0, 0, 0), (384, 384, 16, 1, 0, 0),
    (192, 192, 32, 0, 1, 0), (192, 384, 16, 0, 0, 1), (384, 384, 16, 0, 0, 1), (576, 384, 16, 0, 0, 1),
    (768, 384, 16, 0, 0, 1), (384, 384, 8, 1, 0, 0), (384, 384, 16, 0, 1, 0), (384, 384, 8, 0, 0, 1),
    (768, 384, 8, 0, 0, 1), (384, 384, 4, 1, 0, 0), (384, 384, 8, 0, 1, 0), (384, 384, 4, 0, 0, 0),
    (384, 384, 4, 0, 0, 1), (768, 384, 4, 0, 0, 0)]


@pytest.mark.parametrize("winograd", [False, True])
@pytest.mark.parametrize("cls", G2_CLASSES, ids=lambda c: "blk_%d_%d_%d_%d%d%d" % c)
def test_unet_block_full_width_vs_reference_golden(gpu, golden_dir, monkeypatch, cls, winograd):
    """SURVEY's G2: each of the 18 UNetBlock classes of the CIFAR network at FULL channel width, forward, input gradient
    and conv1 weight-gradient norm, against the vectors the imported reference produced (tests/golden/g2_blocks.npz).
    winograd=True forces every eligible 3x3 conv (forward, data gradient AND weight gradient) through the Winograd kernels,
    which at the fixture's B=1 would otherwise stay on the direct kernels."""
    from adm_amd import ops
    from adm_amd.unet.dhariwal import UNetBlock
    monkeypatch.setattr(ops, "WINOGRAD", winograd)
    monkeypatch.setattr(ops, "WINO_MIN_M", 1)
    g = np.load(os.path.join(golden_dir, "g2_blocks.npz"))
    cin, cout, hin, up, down, attn = cls
    name = "blk_%d_%d_%d_%d%d%d" % cls
    import math
    init = dict(init_mode="kaiming_uniform", init_weight=math.sqrt(1 / 3), init_bias=math.sqrt(1 / 3))
    blk = UNetBlock(in_channels=cin, out_channels=cout, emb_channels=768, up=bool(up), down=bool(down), attention=bool(attn),
                    channels_per_head=64, dropout=0.0, init=init,
                    init_zero=dict(init_mode="kaiming_uniform", init_weight=0, init_bias=0)).eval()
    blk.load_state_dict({k: fill.fill_value(name + "." + k, tuple(v.shape)) for k, v in blk.state_dict().items()})
    blk = blk.to(gpu)
    x = fill.hash_tensor((1, cin, hin, hin), name + ".x", 1.0)
    emb = fill.hash_tensor((1, 768), name + ".emb", 1.0)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu).requires_grad_(True)
    y = blk(xd, emb.to(gpu))
    y_nchw = y.permute(0, 3, 1, 2)
    close(y_nchw.reshape(-1)[::7], g[name + ".y"])
    gy = fill.hash_tensor(tuple(y_nchw.shape), name + ".gy", 1.0)
    (y * gy.permute(0, 2, 3, 1).contiguous().to(gpu)).sum().backward()
    close(xd.grad.permute(0, 3, 1, 2).reshape(-1)[::7], g[name + ".dx"])
    want = float(g[name + ".dconv1_norm"])
    got = float(blk.conv1.weight.grad.double().norm())
    assert abs(got - want) <= 1e-3 * want, (got, want)


def make_ddpm(sched, gpu):
    import importlib
    variant, eps, smin = ("uncond_unet", 1e-4, 0.01) if sched == "const" else ("uncond_unet_sd_2", 1e-3, 0.001)
    m, cfg, sd = build_unet(variant, gpu)
    D = importlib.import_module("ddm.ddm_" + sched).DDPM
    mcfg = dict(eps=eps, sigma_max=1, sigma_min=smin, weighting_loss=True, use_augment=False)
    dpm = D(model=m, image_size=[32, 32], sampling_timesteps=10, loss_type="l2", start_dist="normal",
            perceptual_weight=0.0, use_l1=False, cfg=mcfg, **mcfg).to(gpu)
    return dpm, cfg, sd, eps, smin


@pytest.mark.parametrize("sched", ["const", "const_2"])
def test_training_step_vs_golden(gpu, golden_dir, sched):
    g6 = np.load(os.path.join(golden_dir, "g6_training_step.npz"))
    dpm, cfg, sd, eps, smin = make_ddpm(sched, gpu)
    dpm.eval()      # dropout is 0 in this config anyway
    x0 = fill.hash_tensor((2, 3, 32, 32), "x0", 1.0)
    noise = fill.hash_tensor((2, 3, 32, 32), "noise", 1.7)
    t = torch.tensor([0.23, 0.81])
    loss, log = dpm.training_step({"image": x0.to(gpu)}, t=t.to(gpu), noise=noise.to(gpu))
    close(loss, g6[sched + ".loss"]); close(log["train/loss_simple"], g6[sched + ".loss_simple"])
    assert set(log) == {"train/loss_simple", "train/loss_vlb", "train/loss"}
    loss.backward()
    grads = [p.grad for p in dpm.parameters() if p.grad is not None]
    gn = torch.sqrt(sum(gr.double().pow(2).sum() for gr in grads))
    close(gn, g6[sched + ".grad_norm"])
    close(dict(dpm.named_parameters())["model.model.map_layer1.bias"].grad, g6[sched + ".grad.map_layer1.bias"],
          scale=float(g6[sched + ".grad_norm"]) / 8)
    if sched == "const":
        close(dpm.q_sample(x0.to(gpu), noise.to(gpu), t.to(gpu)), g6["const.x_noisy"])


@pytest.mark.parametrize("sched", ["const", "const_2"])
def test_deterministic_sampler_vs_golden(gpu, golden_dir, sched):
    g7 = np.load(os.path.join(golden_dir, "g7_samplers.npz"))
    dpm, cfg, sd, eps, smin = make_ddpm(sched, gpu)
    dpm.eval()
    xT = fill.hash_tensor((2, 3, 32, 32), "xT", 1.7, torch.float64)
    img, traj = dpm.sample_fn_d((2, 3, 32, 32), x_T=xT.to(gpu), return_traj=True)
    assert img.dtype == torch.float64 and float(img.min()) >= 0 and float(img.max()) <= 1
    close(traj[2], g7[sched + ".x_after_step3"])
    close(img, g7[sched + ".img"])
    close(dpm.t_steps(), g7[sched + ".t_steps"], rtol=1e-12, atol=0)
    img2 = dpm.sample(batch_size=2, x_T=xT.to(gpu))
    assert torch.equal(img, img2)
    if sched == "const_2":
        draws = [fill.hash_tensor((2, 3, 32, 32), f"s{k}", 1.7, torch.float64) for k in range(11)]
        simg = dpm.sample_fn_s((2, 3, 32, 32), x_T=draws[0].to(gpu), epsilons=draws[1:])
        close(simg, g7["const_2.stochastic_img"])


def test_stochastic_sampler_const_vs_oracle(gpu):
    """sample_fn_s for the sqrt(t) schedule (ddm_const.py:380-422; that module cannot be imported, so the pin is
    the oracle restatement run on the CPU with the same injected draws)."""
    dpm, cfg, sd, eps, smin = make_ddpm("const", gpu)
    dpm.eval()
    draws = [fill.hash_tensor((2, 3, 32, 32), f"s{k}", 1.7, torch.float64) for k in range(11)]
    img = dpm.sample_fn_s((2, 3, 32, 32), x_T=draws[0].to(gpu), epsilons=draws[1:])
    assert img.dtype == torch.float64 and float(img.min()) >= 0 and float(img.max()) <= 1
    with torch.no_grad():
        want = ddm_ref.sample_fn_s("const", lambda x, tt: unet_ref.edm_precond(sd, cfg, x, tt), draws[0], draws[1:], 10,
                                   smin, 1.0)
    close(img, want)
    dpm.cfg["sample_type"] = "stochastic"
    torch.manual_seed(3)
    a = dpm.sample(batch_size=2)
    assert a.shape == (2, 3, 32, 32) and float(a.min()) >= 0 and float(a.max()) <= 1


def test_training_mode_dropout_runs_and_is_seeded(gpu):
    m, cfg, _ = build_unet("uncond_unet", gpu, dropout=0.1)
    m.train()
    x, sigma, _ = small_inputs(cfg)
    torch.manual_seed(7)
    a = m(x.to(gpu), sigma.to(gpu))[0]
    b = m(x.to(gpu), sigma.to(gpu))[0]
    assert not torch.equal(a, b)            # fresh mask per call
    m.eval()
    c = m(x.to(gpu), sigma.to(gpu))[0]
    d = m(x.to(gpu), sigma.to(gpu))[0]
    assert torch.equal(c, d)


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path must fail loudly off-GPU."""
    from adm_amd import ops
    with pytest.raises(RuntimeError):
        ops.silu(torch.zeros(4))


def test_latent_unet_config3_vs_oracle(gpu):
    """BASELINE configs[3]'s UNet (configs/celebahq/celeb_uncond_ddm_const2_unet_ldm.yaml:42-55 in the reference):
    single-decoder uncond_unet_sd_2 on 64x64x3 latents, model_channels=128, attention at 16x16 / 8x8 -- forward and
    gradients against the CPU oracle (the KL-f4 autoencoder around it is not part of this build)."""
    import importlib
    cfg = unet_ref.default_cfg(variant="uncond_unet_sd_2", img_resolution=64, model_channels=128, num_blocks=1,
                               attn_resolutions=[16, 8], dropout=0.0, augment_dim=0)
    mod = importlib.import_module("unet.uncond_unet_sd_2")
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions",
                              "dropout", "augment_dim")}
    m = mod.EDMPrecond(img_resolution=64, img_channels=3, model_type="DhariwalUNet", **kw)
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu).eval()
    x = fill.hash_tensor((2, 3, 64, 64), "z", 1.0)
    sigma = torch.tensor([0.3, 0.9])
    dx, dy = m(x.to(gpu), sigma.to(gpu))
    sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd.items()}
    ox, oy = unet_ref.edm_precond(sdo, cfg, x, sigma)
    close(dx, ox.detach()); close(dy, oy.detach())
    gx = fill.hash_tensor(dx.shape, "gz", 1.0)
    ((dx * gx.to(gpu)).sum() + (dy * gx.to(gpu)).sum()).backward()
    ((ox * gx).sum() + (oy * gx).sum()).backward()
    gmax = max(float(v.grad.double().norm()) for v in sdo.values() if v.grad is not None)
    bad = []
    for name, p in m.named_parameters():
        want = sdo[name].grad
        err = float((p.grad.cpu().double() - want.double()).norm() / (want.double().norm() + 1e-6 * gmax))
        if err > 2e-3:
            bad.append((name, err))
    assert not bad, bad[:10]


def test_latent_unet_with_32x32_attention_vs_oracle(gpu):
    """celeb_uncond_ddm_const_uncond_unet_ldm.yaml:42-55: attn_resolutions [32, 16] on 64x64 latents -> L = 1024
    attention through the chunked (online-softmax) kernels; two-decoder variant, reduced depth."""
    import importlib
    cfg = unet_ref.default_cfg(variant="uncond_unet", img_resolution=64, model_channels=64, num_blocks=1,
                               attn_resolutions=[32, 16], dropout=0.0, augment_dim=0)
    mod = importlib.import_module("unet.uncond_unet")
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions",
                              "dropout", "augment_dim")}
    m = mod.EDMPrecond(img_resolution=64, img_channels=3, model_type="DhariwalUNet", **kw)
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu).eval()
    x = fill.hash_tensor((2, 3, 64, 64), "z", 1.0)
    sigma = torch.tensor([0.3, 0.9])
    dx, dy = m(x.to(gpu), sigma.to(gpu))
    sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd.items()}
    ox, oy = unet_ref.edm_precond(sdo, cfg, x, sigma)
    close(dx, ox.detach()); close(dy, oy.detach())
    gx = fill.hash_tensor(dx.shape, "gz", 1.0)
    ((dx * gx.to(gpu)).sum() + (dy * gx.to(gpu)).sum()).backward()
    ((ox * gx).sum() + (oy * gx).sum()).backward()
    gmax = max(float(v.grad.double().norm()) for v in sdo.values() if v.grad is not None)
    bad = []
    for name, p in m.named_parameters():
        want = sdo[name].grad
        err = float((p.grad.cpu().double() - want.double()).norm() / (want.double().norm() + 1e-6 * gmax))
        if err > 2e-3:
            bad.append((name, err))
    assert not bad, bad[:10]


def test_sampler_hip_graph_replay_equals_eager(gpu, monkeypatch):
    """ADM_SAMPLE_GRAPH=1: the 10 UNet forwards replay from one captured HIP graph; results must be bit-identical to the
    eager launches, also on a second call (cached graph) and after a weight update (graph re-captured)."""
    from adm_amd.ddm.ddm_const import DDPM
    from adm_amd import ops
    m, cfg, sd = build_unet("uncond_unet", gpu)
    mcfg = dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True)
    dpm = DDPM(model=m, image_size=[32, 32], sampling_timesteps=10, perceptual_weight=0.0, cfg=mcfg).to(gpu).eval()
    xT = fill.hash_tensor((3, 3, 32, 32), "xTg", 1.7, torch.float64).to(gpu)
    eager = dpm.sample(batch_size=3, x_T=xT)
    monkeypatch.setenv("ADM_SAMPLE_GRAPH", "1")
    g1 = dpm.sample(batch_size=3, x_T=xT)
    g1_copy = g1.clone()
    g2 = dpm.sample(batch_size=3, x_T=xT * 0.5)
    # the returned tensor must not alias the graph's static input buffer -- a second call of the same shape
    # would overwrite the first result in place
    assert g1.data_ptr() != g2.data_ptr() and torch.equal(g1, g1_copy) and not torch.equal(g1, g2)
    g2 = dpm.sample(batch_size=3, x_T=xT)
    assert torch.equal(eager, g1) and torch.equal(eager, g2)
    assert len(dpm._graphs) == 1
    with torch.no_grad():
        dict(dpm.named_parameters())["model.model.out_conv.weight"].mul_(1.5)
    ops.invalidate_packed()
    g3 = dpm.sample(batch_size=3, x_T=xT)
    monkeypatch.setenv("ADM_SAMPLE_GRAPH", "0")
    e3 = dpm.sample(batch_size=3, x_T=xT)
    assert torch.equal(g3, e3) and not torch.equal(g3, eager)


def test_affine_group_equals_per_block_linears(gpu, monkeypatch):
    """ops.affine_group: the `affine` Linears of all blocks as one GEMM (their scale/shift read in place as column slices, the
    GroupNorm backward writing d scale/shift into one buffer, two launches for all their gradients) against the per-block Linears:
    same outputs, same gradients for EVERY parameter -- the Linears', the embedding MLP's behind them -- and for the input; then
    once more after the parameters changed in place and after a load_state_dict (the re-homed operands must follow)."""
    from adm_amd import ops
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(ops, "AFFINE_GROUP", mode)
        m, cfg, sd = build_unet("uncond_unet", gpu)
        m.train()
        x, sigma, aug = small_inputs(cfg)
        outs = []
        for rnd in range(3):
            if rnd == 1:
                with torch.no_grad():
                    for n, p in m.named_parameters():
                        if "affine" in n:
                            p.mul_(1.25).add_(0.01)
                ops.invalidate_packed()
            if rnd == 2:
                m.load_state_dict({k: (v * 0.5 if "affine" in k else v) for k, v in sd.items()})
            m.zero_grad()
            xg = x.to(gpu).requires_grad_(True)
            dx, dy = m(xg, sigma.to(gpu), augment_labels=aug.to(gpu))
            gx, gy = fill.hash_tensor(dx.shape, "gx", 1.0).to(gpu), fill.hash_tensor(dy.shape, "gy", 1.0).to(gpu)
            ((dx * gx).sum() + (dy * gy).sum()).backward()
            outs.append((dx.detach().clone(), dy.detach().clone(), xg.grad.clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
        with torch.no_grad():        # the sampler's call pattern: 0-dim sigma -> one embedding row for the whole batch
            s0 = m(x.to(gpu).double(), torch.tensor(0.37, dtype=torch.float64, device=gpu))
        res[mode] = (outs, s0)
    for rnd in range(3):
        a, b = res[False][0][rnd], res[True][0][rnd]
        close(b[0], a[0], rtol=1e-5, atol=1e-6); close(b[1], a[1], rtol=1e-5, atol=1e-6)
        close(b[2], a[2], rtol=1e-4, atol=1e-5)
        gmax = max(float(g.double().norm()) for g in a[3].values())
        for n, g in a[3].items():
            err = float((b[3][n].double() - g.double()).norm() / (g.double().norm() + 1e-6 * gmax))
            assert err <= (2e-4 if ("affine" in n or "map_" in n) else 1e-3), (rnd, n, err)
    assert not torch.equal(res[True][0][0][0], res[True][0][1][0]) and not torch.equal(res[True][0][1][0], res[True][0][2][0])
    close(res[True][1][0], res[False][1][0], rtol=1e-5, atol=1e-6)


def test_fp16_split_format_in_the_training_step(gpu, monkeypatch):
    """The fp16 split format end to end on the reduced two-decoder UNet (every 3x3 layer forced onto the Winograd kernels): the forward
    convs take their bound from the GroupNorm that feeds them, the data-gradient convs from the GroupNorm backward / gradient sum /
    concatenation split that produced their dy (ops._reg_amax / _get_amax, by address); with ADM_AMAX_CHECK semantics on, EVERY bound
    that reaches a conv is verified against the tensor it came with.  Outputs and all parameter gradients equal the bf16-format run."""
    from adm_amd import ops
    monkeypatch.setattr(ops, "WINO_MIN_M", 1)
    res = {}
    for h3 in (False, True):
        monkeypatch.setattr(ops, "FP16X3", h3)
        monkeypatch.setattr(ops, "AMAX_CHECK", h3)
        m, cfg, _ = build_unet("uncond_unet", gpu)
        m.train()
        x, sigma, aug = small_inputs(cfg)
        monkeypatch.setattr(ops, "PROFILE", [])
        dx, dy = m(x.to(gpu), sigma.to(gpu), augment_labels=aug.to(gpu))
        gx, gy = fill.hash_tensor(dx.shape, "gx", 1.0).to(gpu), fill.hash_tensor(dy.shape, "gy", 1.0).to(gpu)
        ((dx * gx).sum() + (dy * gy).sum()).backward()
        tags = [r[4] for r in ops.PROFILE if r[0] == "wino2h3"]
        monkeypatch.setattr(ops, "PROFILE", None)
        res[h3] = (dx.detach().clone(), dy.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}, tags)
    tags = res[True][3]
    n_fwd, n_dgrad = sum(t.startswith("fwd") for t in tags), sum(t.startswith("dgrad") for t in tags)
    assert not res[False][3] and n_fwd >= 40 and n_dgrad >= 40, (n_fwd, n_dgrad)
    close(res[True][0], res[False][0], rtol=1e-5, atol=1e-5); close(res[True][1], res[False][1], rtol=1e-5, atol=1e-5)
    gmax = max(float(g.double().norm()) for g in res[False][2].values())
    for n, g in res[False][2].items():
        err = float((res[True][2][n].double() - g.double()).norm() / (g.double().norm() + 1e-6 * gmax))
        assert err <= 2e-4, (n, err)
