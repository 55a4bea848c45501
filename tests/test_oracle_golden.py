"""CPU: the oracle restatement vs the committed golden vectors the imported reference produced
(tools/make_golden.py).  rtol 1e-3 / atol 1e-4 is north_star's fp32 tolerance; the oracle uses the
same ATen ops as the reference so it actually matches to ~1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import ddm_ref, fill, unet_ref

SMALL = dict(model_channels=64, num_blocks=1, dropout=0.0)
RTOL, ATOL = 1e-3, 1e-4


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    np.testing.assert_allclose(a, np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


def small_inputs(cfg, B=2, tag="x"):
    x = fill.hash_tensor((B, 3, 32, 32), tag, 1.0)
    sigma = torch.tensor([0.05, 0.7, 0.31, 0.999][:B])
    aug = fill.hash_tensor((B, cfg["augment_dim"]), tag + "aug", 1.0)
    return x, sigma, aug


@pytest.mark.parametrize("variant", unet_ref.VARIANTS)
@pytest.mark.parametrize("use_aug", [0, 1])
def test_precond_small_forward_and_grads(golden_dir, variant, use_aug):
    g = np.load(os.path.join(golden_dir, "g5_precond_small.npz"))
    cfg = unet_ref.default_cfg(variant=variant, **SMALL)
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    sd = {k: v.requires_grad_("resample" not in k) for k, v in sd.items()}
    x, sigma, aug = small_inputs(cfg)
    x.requires_grad_(True)
    dx, dy = unet_ref.edm_precond(sd, cfg, x, sigma, **(dict(augment_labels=aug) if use_aug else {}))
    p = f"{variant}.aug{use_aug}."
    close(dx, g[p + "D_x"]); close(dy, g[p + "D_y"])
    ((dx * fill.hash_tensor(dx.shape, "gx", 1.0)).sum() + (dy * fill.hash_tensor(dy.shape, "gy", 1.0)).sum()).backward()
    close(x.grad, g[p + "dL_dx"])
    for key in g.files:
        if key.startswith(p + "grad."):
            name = key[len(p + "grad."):]
            close(sd[name].grad.reshape(-1)[:4096], g[key], atol=1e-4 * float(g[p + "gradnorm." + name]) + 1e-6)


def test_scalar_fp64_sigma(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5b_scalar_sigma.npz"))
    cfg = unet_ref.default_cfg(variant="uncond_unet", **SMALL)
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    x, _, _ = small_inputs(cfg)
    with torch.no_grad():
        dx, dy = unet_ref.edm_precond(sd, cfg, x.double(), torch.tensor(0.37, dtype=torch.float64))
    assert dx.dtype == torch.float32
    close(dx, g["D_x"]); close(dy, g["D_y"])


def test_param_census_full_width():
    shapes = unet_ref.param_shapes(unet_ref.default_cfg())
    n = sum(int(np.prod(s)) for k, s in shapes.items() if "resample" not in k)
    assert n == 216141136                       # SURVEY.md section 2.2 [measured on the reference]
    assert len(shapes) == 829
    blocks = unet_ref.block_specs(unet_ref.default_cfg())
    nblk = sum(b["kind"] == "block" for b in blocks["enc"]) + 2 * len(blocks["dec"])
    assert nblk == 57


def test_primitives(golden_dir):
    g = np.load(os.path.join(golden_dir, "g134_primitives.npz"))
    for C, H in ((192, 32), (384, 16), (576, 8), (768, 4), (64, 8), (96, 8)):
        w, b = fill.fill_value(f"gn{C}.norm0.weight", (C,)), fill.fill_value(f"gn{C}.norm0.bias", (C,))
        x = fill.hash_tensor((2, C, H, H), f"gn{C}.x", 2.0) + 0.3
        y = torch.nn.functional.silu(unet_ref._gn({"p.weight": w, "p.bias": b}, "p", x))
        close(y.reshape(-1)[::5], g[f"gn_silu.C{C}.H{H}"])
    for L, heads in ((16, 6), (64, 6), (256, 6), (64, 1)):
        h = int(L ** 0.5)
        qkv = fill.hash_tensor((2, 3 * 64 * heads, h, h), f"attn{L}.{heads}", 1.5)
        close(unet_ref.attention_core(qkv, heads).reshape(-1)[::3], g[f"attn.L{L}.h{heads}"])


def test_blocks_small_classes(golden_dir):
    """The 4x4 / 8x8 block classes of SURVEY 2.2 at full channel width (the big ones are checked
    at generation time; see oracle_vs_reference_report.json)."""
    g = np.load(os.path.join(golden_dir, "g2_blocks.npz"))
    for (cin, cout, hin, up, down, attn) in [(384, 384, 4, 0, 0, 1), (768, 384, 4, 0, 0, 0), (384, 384, 4, 1, 0, 0),
                                             (384, 384, 8, 0, 1, 0), (768, 384, 8, 0, 0, 1)]:
        name = f"blk_{cin}_{cout}_{hin}_{up}{down}{attn}"
        b = dict(cin=cin, cout=cout, up=bool(up), down=bool(down), attn=bool(attn), res=hin)
        shapes = {}
        emb = 768
        shapes[name + ".norm0.weight"] = (cin,); shapes[name + ".norm0.bias"] = (cin,)
        shapes[name + ".conv0.weight"] = (cout, cin, 3, 3); shapes[name + ".conv0.bias"] = (cout,)
        shapes[name + ".affine.weight"] = (2 * cout, emb); shapes[name + ".affine.bias"] = (2 * cout,)
        shapes[name + ".norm1.weight"] = (cout,); shapes[name + ".norm1.bias"] = (cout,)
        shapes[name + ".conv1.weight"] = (cout, cout, 3, 3); shapes[name + ".conv1.bias"] = (cout,)
        if cin != cout:
            shapes[name + ".skip.weight"] = (cout, cin, 1, 1); shapes[name + ".skip.bias"] = (cout,)
        if attn:
            shapes[name + ".norm2.weight"] = (cout,); shapes[name + ".norm2.bias"] = (cout,)
            shapes[name + ".qkv.weight"] = (3 * cout, cout, 1, 1); shapes[name + ".qkv.bias"] = (3 * cout,)
            shapes[name + ".proj.weight"] = (cout, cout, 1, 1); shapes[name + ".proj.bias"] = (cout,)
        sd = {k: fill.fill_value(k, s) for k, s in shapes.items()}
        x = fill.hash_tensor((1, cin, hin, hin), name + ".x", 1.0).requires_grad_(True)
        e = fill.hash_tensor((1, emb), name + ".emb", 1.0)
        y = unet_ref.unet_block(sd, name, b, x, e)
        close(y.reshape(-1)[::7], g[name + ".y"])
        (y * fill.hash_tensor(y.shape, name + ".gy", 1.0)).sum().backward()
        close(x.grad.reshape(-1)[::7], g[name + ".dx"])


def test_training_step_and_samplers(golden_dir):
    g6 = np.load(os.path.join(golden_dir, "g6_training_step.npz"))
    g7 = np.load(os.path.join(golden_dir, "g7_samplers.npz"))
    x0 = fill.hash_tensor((2, 3, 32, 32), "x0", 1.0)
    noise = fill.hash_tensor((2, 3, 32, 32), "noise", 1.7)
    t = torch.tensor([0.23, 0.81])
    xT = fill.hash_tensor((2, 3, 32, 32), "xT", 1.7, torch.float64)
    for sched, variant, eps, smin in (("const", "uncond_unet", 1e-4, 0.01), ("const_2", "uncond_unet_sd_2", 1e-3, 0.001)):
        cfg = unet_ref.default_cfg(variant=variant, **SMALL)
        sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
        sd = {k: v.requires_grad_("resample" not in k) for k, v in sd.items()}
        mf = lambda x, tt, **kw: unet_ref.edm_precond(sd, cfg, x, tt, **kw)
        loss, log, _ = ddm_ref.p_losses(sched, mf, x0, t, noise, eps, True)
        close(loss, g6[sched + ".loss"]); close(log["train/loss_simple"], g6[sched + ".loss_simple"])
        loss.backward()
        gn = torch.sqrt(sum(v.grad.double().pow(2).sum() for v in sd.values() if v.grad is not None))
        close(gn, g6[sched + ".grad_norm"])
        close(sd["model.map_layer1.bias"].grad, g6[sched + ".grad.map_layer1.bias"], atol=1e-4 * float(gn))
        with torch.no_grad():
            img, traj = ddm_ref.sample_fn_d(sched, mf, xT, 10, smin, 1.0, return_traj=True)
        assert img.dtype == torch.float64 and float(img.min()) >= 0 and float(img.max()) <= 1
        close(img, g7[sched + ".img"]); close(traj[2], g7[sched + ".x_after_step3"])
        close(ddm_ref.t_steps_deterministic(sched, 10, smin, 1.0), g7[sched + ".t_steps"], rtol=1e-12, atol=0)
    draws = [fill.hash_tensor((2, 3, 32, 32), f"s{k}", 1.7, torch.float64) for k in range(11)]
    with torch.no_grad():
        img = ddm_ref.sample_fn_s("const_2", mf, draws[0], draws[1:], 10, 0.001, 1.0)
    close(img, g7["const_2.stochastic_img"])


def test_known_answers_and_schedules(golden_dir):
    one = lambda v: torch.full((1, 1, 1, 1), v)
    xt = ddm_ref.q_sample("const", one(0.5), one(-1.0), torch.tensor([0.25]), one(-0.5))
    assert abs(float(xt) + 0.125) < 1e-7
    assert abs(float(ddm_ref.pred_x0_from_xt("const", xt, one(-1.0), one(-0.5), torch.tensor([0.25]))) - 0.5) < 1e-7
    ts = ddm_ref.t_steps_deterministic("const", 10, 0.01, 1.0)
    assert abs(float(ts[1]) - 0.88890) < 1e-12 and abs(float(ts[9]) - 1e-4) < 1e-15 and float(ts[10]) == 0.0
    g9 = np.load(os.path.join(golden_dir, "g9_schedules.npz"))
    for s, d in zip(g9["ema_steps"], g9["ema_decay"]):
        assert abs(ddm_ref.ema_decay(int(s)) - d) < 1e-12
    for i, r in zip(g9["lr_its"], g9["lr_ratio"]):
        assert abs(ddm_ref.lr_lambda(int(i), 1e-4, 5e-6, 800000) - r) < 1e-12
