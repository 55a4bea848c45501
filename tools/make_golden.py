#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

1. Imports the reference's own modules (unet.uncond_unet*, ddm.ddm_const_2, ddm.ema) -- with the two
   in-process shims SURVEY.md section 8c describes (an 'ADM' package alias and an empty
   'torchvision' stub, both only touched at import time) -- and checks the oracle restatement in
   ``oracle/`` against them on identical inputs, at reduced AND full width.
2. Writes compact fixtures (inputs are closed-form, see oracle/fill.py, so only expected outputs
   are stored) to tests/golden/*.npz and a human-readable report to
   tests/golden/oracle_vs_reference_report.json.

The fixtures are data (tensors the reference produced); no reference source is copied.
Usage:  python tools/make_golden.py
"""
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# --- import-time shims ---------------------------------------------------
adm = types.ModuleType("ADM"); adm.__path__ = [REF]; sys.modules["ADM"] = adm
tv = types.ModuleType("torchvision"); tv.models = types.ModuleType("torchvision.models")
tv.transforms = types.ModuleType("torchvision.transforms")
sys.modules["torchvision"] = tv; sys.modules["torchvision.models"] = tv.models
sys.modules["torchvision.transforms"] = tv.transforms

import importlib  # noqa: E402

from oracle import ddm_ref, fill, unet_ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)
report = {"torch": torch.__version__, "cases": []}


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(name, got, want, tol=2e-5):
    e = rel_err(got, want)
    ok = e <= tol
    report["cases"].append(dict(case=name, max_rel_err=e, tol=tol, ok=bool(ok)))
    print(f"{'OK ' if ok else 'BAD'} {name}: rel_err={e:.3e}")
    assert ok, name
    return e


SMALL = dict(model_channels=64, num_blocks=1, dropout=0.0)
MODS = {v: importlib.import_module("unet." + v) for v in unet_ref.VARIANTS}


def build_ref_unet(cfg):
    mod = MODS[cfg["variant"]]
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks",
                              "attn_resolutions", "dropout", "augment_dim")}
    m = mod.EDMPrecond(img_resolution=cfg["img_resolution"], img_channels=cfg["img_channels"],
                       model_type="DhariwalUNet", **kw)
    shapes = unet_ref.param_shapes(cfg)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "state_dict names/order differ"
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = fill.filled_state_dict(shapes)
    m.load_state_dict(sd, strict=True)
    return m.eval(), sd


def inputs(cfg, B, tag="x"):
    R, C = cfg["img_resolution"], cfg["img_channels"]
    x = fill.hash_tensor((B, C, R, R), tag, 1.0)
    sigma = torch.tensor([0.05, 0.7, 0.31, 0.999, 1e-4, 0.5][:B] if B <= 6 else
                         (fill.hash_tensor((B,), tag + "s", 0.49) + 0.5).tolist(), dtype=torch.float32)
    aug = fill.hash_tensor((B, cfg["augment_dim"]), tag + "aug", 1.0)
    return x, sigma, aug


# ------------------------------------------------------------------------------------------------
# G5: EDMPrecond end-to-end, all five variants, reduced width: forward + gradients
# ------------------------------------------------------------------------------------------------
GRAD_KEYS = ["model.map_layer0.weight", "model.enc.32x32_conv.weight", "model.enc.16x16_block0.qkv.weight",
             "model.enc.8x8_down.conv0.weight", "model.dec.4x4_in0.proj.weight", "model.dec.16x16_up.conv0.weight",
             "model.dec.32x32_block1.skip.weight", "model.dec.8x8_block0.norm1.weight", "model.decouple1.1.map.weight",
             "model.decouple1.1.q_conv.weight", "model.dec.32x32_block0.affine.bias", "model.out_conv.weight",
             "model.map_augment.weight"]

g5 = {}
for variant in unet_ref.VARIANTS:
    cfg = unet_ref.default_cfg(variant=variant, **SMALL)
    m, sd = build_ref_unet(cfg)
    x, sigma, aug = inputs(cfg, 2)
    for use_aug in (False, True):
        kw = dict(augment_labels=aug) if use_aug else {}
        xr = x.clone().requires_grad_(True)
        dx_ref, dy_ref = m(xr, sigma, **kw)
        wsum = lambda a, b: (a * fill.hash_tensor(a.shape, "gx", 1.0)).sum() + (b * fill.hash_tensor(b.shape, "gy", 1.0)).sum()
        wsum(dx_ref, dy_ref).backward()
        sdo = {k: v.clone().requires_grad_(v.is_floating_point() and "resample" not in k) for k, v in sd.items()}
        xo = x.clone().requires_grad_(True)
        dx_o, dy_o = unet_ref.edm_precond(sdo, cfg, xo, sigma, **kw)
        wsum(dx_o, dy_o).backward()
        tag = f"G5/{variant}/aug{int(use_aug)}"
        check(tag + "/D_x", dx_o, dx_ref); check(tag + "/D_y", dy_o, dy_ref)
        check(tag + "/dL_dx", xo.grad, xr.grad, 1e-4)
        named = dict(m.named_parameters())
        gk = [k for k in GRAD_KEYS if k in named and (use_aug or "map_augment" not in k)]
        for k in gk:
            check(tag + "/grad/" + k, sdo[k].grad, named[k].grad, 1e-4)
        g5[f"{variant}.aug{int(use_aug)}.D_x"] = dx_ref.detach().numpy()
        g5[f"{variant}.aug{int(use_aug)}.D_y"] = dy_ref.detach().numpy()
        g5[f"{variant}.aug{int(use_aug)}.dL_dx"] = xr.grad.numpy()
        for k in gk:
            g = named[k].grad
            g5[f"{variant}.aug{int(use_aug)}.grad.{k}"] = g.reshape(-1)[:4096].numpy().copy()
            g5[f"{variant}.aug{int(use_aug)}.gradnorm.{k}"] = np.array(float(g.double().norm()))
        m.zero_grad()
np.savez_compressed(os.path.join(OUT, "g5_precond_small.npz"), **g5)

# G5b: sigma as a 0-dim fp64 tensor and x fp64 (the sampling call pattern), reduced width
cfg = unet_ref.default_cfg(variant="uncond_unet", **SMALL)
m, sd = build_ref_unet(cfg)
x, _, _ = inputs(cfg, 2)
with torch.no_grad():
    a_ref = m(x.double(), torch.tensor(0.37, dtype=torch.float64))
    a_o = unet_ref.edm_precond(sd, cfg, x.double(), torch.tensor(0.37, dtype=torch.float64))
check("G5b/scalar_sigma/D_x", a_o[0], a_ref[0]); check("G5b/scalar_sigma/D_y", a_o[1], a_ref[1])
np.savez_compressed(os.path.join(OUT, "g5b_scalar_sigma.npz"), D_x=a_ref[0].numpy(), D_y=a_ref[1].numpy())

# ------------------------------------------------------------------------------------------------
# G5c: FULL-WIDTH CIFAR-10 config (216 M params), B=1, forward
# ------------------------------------------------------------------------------------------------
cfg_full = unet_ref.default_cfg(variant="uncond_unet", dropout=0.0)
t0 = time.time()
m_full, sd_full = build_ref_unet(cfg_full)
nparam = sum(p.numel() for p in m_full.parameters())
report["full_params"] = nparam
assert nparam == 216141136, nparam
x, sigma, aug = inputs(cfg_full, 1)
with torch.no_grad():
    f_ref = m_full(x, sigma, augment_labels=aug)
    f_o = unet_ref.edm_precond(sd_full, cfg_full, x, sigma, augment_labels=aug)
check("G5c/full_width/D_x", f_o[0], f_ref[0]); check("G5c/full_width/D_y", f_o[1], f_ref[1])
np.savez_compressed(os.path.join(OUT, "g5c_full_width.npz"), D_x=f_ref[0].numpy(), D_y=f_ref[1].numpy())
print(f"full-width case took {time.time() - t0:.1f}s")

# ------------------------------------------------------------------------------------------------
# G2: UNetBlock classes of SURVEY section 2.2 at FULL channel width, B=1 -- checked here, outputs
#     stored strided (every 7th element) to keep fixtures small.
# ------------------------------------------------------------------------------------------------
U = MODS["uncond_unet"]
BLOCK_CLASSES = [  # cin, cout, Hin, up, down, attn
    (192, 192, 32, 0, 0, 0), (384, 192, 32, 0, 0, 0), (576, 192, 32, 0, 0, 0), (384, 384, 16, 1, 0, 0),
    (192, 192, 32, 0, 1, 0), (192, 384, 16, 0, 0, 1), (384, 384, 16, 0, 0, 1), (576, 384, 16, 0, 0, 1),
    (768, 384, 16, 0, 0, 1), (384, 384, 8, 1, 0, 0), (384, 384, 16, 0, 1, 0), (384, 384, 8, 0, 0, 1),
    (768, 384, 8, 0, 0, 1), (384, 384, 4, 1, 0, 0), (384, 384, 8, 0, 1, 0), (384, 384, 4, 0, 0, 0),
    (384, 384, 4, 0, 0, 1), (768, 384, 4, 0, 0, 0)]
g2 = {}
init = dict(init_mode="kaiming_uniform", init_weight=np.sqrt(1 / 3), init_bias=np.sqrt(1 / 3))
for (cin, cout, hin, up, down, attn) in BLOCK_CLASSES:
    blk = U.UNetBlock(in_channels=cin, out_channels=cout, emb_channels=768, up=bool(up), down=bool(down),
                      attention=bool(attn), channels_per_head=64, dropout=0.0, init=init,
                      init_zero=dict(init_mode="kaiming_uniform", init_weight=0, init_bias=0)).eval()
    name = f"blk_{cin}_{cout}_{hin}_{up}{down}{attn}"
    sd = {name + "." + k: fill.fill_value(name + "." + k, tuple(v.shape)) for k, v in blk.state_dict().items()}
    blk.load_state_dict({k[len(name) + 1:]: v for k, v in sd.items()})
    x = fill.hash_tensor((1, cin, hin, hin), name + ".x", 1.0).requires_grad_(True)
    emb = fill.hash_tensor((1, 768), name + ".emb", 1.0)
    y_ref = blk(x, emb)
    gw = fill.hash_tensor(y_ref.shape, name + ".gy", 1.0)
    (y_ref * gw).sum().backward()
    sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd.items()}
    xo = x.detach().clone().requires_grad_(True)
    bdesc = dict(cin=cin, cout=cout, up=bool(up), down=bool(down), attn=bool(attn))
    y_o = unet_ref.unet_block(sdo, name, bdesc, xo, emb)
    (y_o * gw).sum().backward()
    check(f"G2/{name}/y", y_o, y_ref); check(f"G2/{name}/dx", xo.grad, x.grad, 1e-4)
    check(f"G2/{name}/dconv0", sdo[name + ".conv0.weight"].grad, blk.conv0.weight.grad, 1e-4)
    check(f"G2/{name}/daffine", sdo[name + ".affine.weight"].grad, blk.affine.weight.grad, 1e-4)
    g2[name + ".y"] = y_ref.detach().reshape(-1)[::7].numpy().copy()
    g2[name + ".dx"] = x.grad.reshape(-1)[::7].numpy().copy()
    g2[name + ".dconv1_norm"] = np.array(float(blk.conv1.weight.grad.double().norm()))
np.savez_compressed(os.path.join(OUT, "g2_blocks.npz"), **g2)

# ------------------------------------------------------------------------------------------------
# G1/G3/G4: GroupNorm+SiLU, attention core (pins the qkv interleave), embedding MLP
# ------------------------------------------------------------------------------------------------
g134 = {}
for C, H in ((192, 32), (384, 16), (576, 8), (768, 4), (64, 8), (96, 8)):
    gnm = U.GroupNorm(C)
    w, b = fill.fill_value(f"gn{C}.norm0.weight", (C,)), fill.fill_value(f"gn{C}.norm0.bias", (C,))
    gnm.load_state_dict(dict(weight=w, bias=b))
    x = fill.hash_tensor((2, C, H, H), f"gn{C}.x", 2.0) + 0.3
    y_ref = torch.nn.functional.silu(gnm(x))
    y_o = torch.nn.functional.silu(unet_ref._gn({"p.weight": w, "p.bias": b}, "p", x))
    check(f"G1/gn_silu/C{C}", y_o, y_ref)
    g134[f"gn_silu.C{C}.H{H}"] = y_ref.detach().reshape(-1)[::5].numpy().copy()
for L, heads in ((16, 6), (64, 6), (256, 6), (64, 1)):
    h = int(L ** 0.5)
    C = 64 * heads
    qkv = fill.hash_tensor((2, 3 * C, h, h), f"attn{L}.{heads}", 1.5)
    q, k, v = qkv.reshape(2 * heads, C // heads, 3, -1).unbind(2)       # reference lines 205-208, verbatim semantics
    wgt = torch.einsum("ncq,nck->nqk", q, k / np.sqrt(k.shape[1])).softmax(dim=2)
    a_ref = torch.einsum("nqk,nck->ncq", wgt, v).reshape(2, C, h, h)
    check(f"G3/attn/L{L}h{heads}", unet_ref.attention_core(qkv, heads), a_ref)
    g134[f"attn.L{L}.h{heads}"] = a_ref.reshape(-1)[::3].numpy().copy()
cfg = unet_ref.default_cfg(variant="uncond_unet", dropout=0.0)
um = m_full.model
for use_aug in (0, 1):
    t = torch.tensor([1e-4, 0.5, 1.0]).log()
    aug = fill.hash_tensor((3, 9), "embaug", 1.0)
    emb = um.map_noise(t)
    if use_aug:
        emb = emb + um.map_augment(aug)
    emb = torch.nn.functional.silu(um.map_layer1(torch.nn.functional.silu(um.map_layer0(emb))))
    emb_o = unet_ref.time_embedding(sd_full, cfg_full, t, aug if use_aug else None)
    check(f"G4/embedding/aug{use_aug}", emb_o, emb)
    g134[f"emb.aug{use_aug}"] = emb.detach().numpy()
np.savez_compressed(os.path.join(OUT, "g134_primitives.npz"), **g134)
del m_full, sd_full

# ------------------------------------------------------------------------------------------------
# G6/G7/G8: diffusion wrapper.  ddm.ddm_const_2 imports (with the shims); ddm.ddm_const does not
# (pytorch_lightning / ldm / cldm absent) so 'const' is pinned by known answers + by running the
# restated formulae through the REAL reference UNet.
# ------------------------------------------------------------------------------------------------
import ddm.ddm_const_2 as D2  # noqa: E402
from ddm.ema import EMA  # noqa: E402

g6 = {}
cfg2 = unet_ref.default_cfg(variant="uncond_unet_sd_2", **SMALL)
m2, sd2 = build_ref_unet(cfg2)
model_cfg = dict(eps=1e-3, sigma_max=1, sigma_min=0.001, weighting_loss=True, use_augment=False)
dpm = D2.DDPM(model=m2, image_size=[32, 32], sampling_timesteps=10, loss_type="l2", start_dist="normal",
              perceptual_weight=0.0, use_l1=False, cfg=dict(model_cfg))
dpm.perceptual_weight = 1.0                       # run the reference p_losses un-modified ...
dpm.perceptual_loss = lambda a, b: torch.zeros(a.shape[0], 1, 1, 1)   # ... with LPIPS == 0 (VGG16 weights not fetchable)
x0 = fill.hash_tensor((2, 3, 32, 32), "x0", 1.0)
noise = fill.hash_tensor((2, 3, 32, 32), "noise", 1.7)
t = torch.tensor([0.23, 0.81])
_orig_randn_like = torch.randn_like
torch.randn_like = lambda *a, **k: noise.clone()
loss_ref, log_ref = dpm.p_losses(x0, t)
torch.randn_like = _orig_randn_like
loss_ref.backward()
sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd2.items()}
mf = lambda x, tt, **kw: unet_ref.edm_precond(sdo, cfg2, x, tt, **kw)
loss_o, log_o, _ = ddm_ref.p_losses("const_2", mf, x0, t, noise, 1e-3, True)
loss_o.backward()
check("G6/const_2/loss", loss_o, loss_ref)
for k in ("train/loss_simple", "train/loss_vlb", "train/loss"):
    check("G6/const_2/" + k, log_o[k], log_ref[k])
gn_ref = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in m2.parameters() if p.grad is not None))
gn_o = torch.sqrt(sum(v.grad.double().pow(2).sum() for v in sdo.values() if v.grad is not None))
check("G6/const_2/grad_norm", gn_o, gn_ref, 1e-4)
g6.update({"const_2.loss": loss_ref.detach().numpy(), "const_2.grad_norm": gn_ref.numpy(),
           "const_2.loss_simple": log_ref["train/loss_simple"].numpy(), "const_2.log_loss": log_ref["train/loss"].numpy(),
           "const_2.grad.map_layer1.bias": dict(m2.named_parameters())["model.map_layer1.bias"].grad.numpy().copy()})

# 'const' through the real two-decoder reference UNet
cfg1 = unet_ref.default_cfg(variant="uncond_unet", **SMALL)
m1, sd1 = build_ref_unet(cfg1)
loss_r, log_r, (xn_r, _, _) = ddm_ref.p_losses("const", lambda x, tt, **kw: m1(x, tt, **kw), x0, t, noise, 1e-4, True)
loss_r.backward()
sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd1.items()}
loss_o, log_o, (xn_o, _, _) = ddm_ref.p_losses("const", lambda x, tt, **kw: unet_ref.edm_precond(sdo, cfg1, x, tt, **kw),
                                               x0, t, noise, 1e-4, True)
loss_o.backward()
check("G6/const/loss(oracle-unet vs reference-unet)", loss_o, loss_r)
gn_r = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in m1.parameters() if p.grad is not None))
gn_o = torch.sqrt(sum(v.grad.double().pow(2).sum() for v in sdo.values() if v.grad is not None))
check("G6/const/grad_norm", gn_o, gn_r, 1e-4)
g6.update({"const.loss": loss_r.detach().numpy(), "const.grad_norm": gn_r.numpy(),
           "const.loss_simple": log_r["train/loss_simple"].numpy(), "const.x_noisy": xn_r.detach().numpy(),
           "const.grad.map_layer1.bias": dict(m1.named_parameters())["model.map_layer1.bias"].grad.numpy().copy()})
# hand-computable known answers: x0=.5, eps=-1, t=.25 -> x_t = .5 - .125 - .5 = -.125
one = lambda v: torch.full((1, 1, 1, 1), v)
ka = ddm_ref.q_sample("const", one(0.5), one(-1.0), torch.tensor([0.25]), one(-0.5))
assert abs(float(ka) - (-0.125)) < 1e-7
assert abs(float(ddm_ref.pred_x0_from_xt("const", ka, one(-1.0), one(-0.5), torch.tensor([0.25]))) - 0.5) < 1e-7
ka2 = ddm_ref.q_sample("const_2", one(0.5), one(-1.0), torch.tensor([0.25]), one(-0.5))
assert abs(float(ka2) - (0.5 - 0.125 - 0.25)) < 1e-7
# reference q_sample / pred_x0 / pred_xtms (const_2) directly
xq = dpm.q_sample(x0, noise, t, -x0)
check("G6/const_2/q_sample", ddm_ref.q_sample("const_2", x0, noise, t, -x0), xq)
check("G6/const_2/pred_x0", ddm_ref.pred_x0_from_xt("const_2", xq, noise, -x0, t), dpm.pred_x0_from_xt(xq, noise, -x0, t))
epsn = fill.hash_tensor((2, 3, 32, 32), "epsn", 1.0, torch.float64)
torch.randn_like = lambda *a, **k: epsn.clone()
xs_ref = dpm.pred_xtms_from_xt(xq, noise, -x0, t, torch.tensor([0.1, 0.3]))
torch.randn_like = _orig_randn_like
check("G6/const_2/pred_xtms", ddm_ref.pred_xtms_from_xt("const_2", xq, noise, -x0, t, torch.tensor([0.1, 0.3]), epsn), xs_ref)
np.savez_compressed(os.path.join(OUT, "g6_training_step.npz"), **g6)

# G7 deterministic sampler trajectories
g7 = {}
xT = fill.hash_tensor((2, 3, 32, 32), "xT", 1.7, torch.float64)
_orig_randn = torch.randn
torch.randn = lambda *a, **k: xT.clone()
img_ref = dpm.sample_fn_d((2, 3, 32, 32))
torch.randn = _orig_randn
with torch.no_grad():
    img_o, traj_o = ddm_ref.sample_fn_d("const_2", lambda x, tt: unet_ref.edm_precond(sd2, cfg2, x, tt), xT, 10, 0.001, 1.0,
                                        return_traj=True)
check("G7/const_2/sample_fn_d", img_o, img_ref, 1e-4)
assert img_ref.dtype == torch.float64
g7["const_2.img"] = img_ref.numpy(); g7["const_2.x_after_step3"] = traj_o[2].numpy()
with torch.no_grad():
    img_r, traj_r = ddm_ref.sample_fn_d("const", lambda x, tt: m1(x, tt), xT, 10, 0.01, 1.0, return_traj=True)
    img_o, traj_o = ddm_ref.sample_fn_d("const", lambda x, tt: unet_ref.edm_precond(sd1, cfg1, x, tt), xT, 10, 0.01, 1.0,
                                        return_traj=True)
check("G7/const/sample_fn_d(oracle-unet vs reference-unet)", img_o, img_r, 1e-4)
ts = ddm_ref.t_steps_deterministic("const", 10, 0.01, 1.0)
assert abs(float(ts[1]) - 0.88890) < 1e-12 and abs(float(ts[9]) - 1e-4) < 1e-15 and float(ts[10]) == 0.0
g7["const.img"] = img_r.numpy(); g7["const.x_after_step3"] = traj_r[2].numpy(); g7["const.t_steps"] = ts.numpy()
g7["const_2.t_steps"] = ddm_ref.t_steps_deterministic("const_2", 10, 0.001, 1.0).numpy()

# G8 stochastic sampler (const_2 through the reference; draws injected in call order)
draws = [fill.hash_tensor((2, 3, 32, 32), f"s{k}", 1.7, torch.float64) for k in range(11)]
it = iter(draws)
torch.randn = lambda *a, **k: next(it).clone()
torch.randn_like = lambda *a, **k: next(it).clone()
dpm.cfg["sample_type"] = "stochastic"
img_ref = dpm.sample_fn_s((2, 3, 32, 32))
torch.randn, torch.randn_like = _orig_randn, _orig_randn_like
with torch.no_grad():
    img_o = ddm_ref.sample_fn_s("const_2", lambda x, tt: unet_ref.edm_precond(sd2, cfg2, x, tt), draws[0], draws[1:], 10,
                                0.001, 1.0)
check("G8/const_2/sample_fn_s", img_o, img_ref, 1e-4)
g7["const_2.stochastic_img"] = img_ref.numpy()
np.savez_compressed(os.path.join(OUT, "g7_samplers.npz"), **g7)

# G9: EMA decay (ddm/ema.py) and the LR lambda (train_uncond_dpm.py:169-177; that file needs fvcore /
# torchvision / tensorboard so the closure is pinned by values computed from its text)
lin = torch.nn.Linear(2, 2)
ema = EMA(lin, beta=0.9996, update_after_step=10000, update_every=8)
g9 = {"ema_steps": np.array([0, 10000, 10001, 10002, 10008, 10080, 20000, 100000, 799999])}
dec = []
for s in g9["ema_steps"]:
    ema.step.fill_(int(s)); dec.append(float(ema.get_current_decay()))
    assert abs(dec[-1] - ddm_ref.ema_decay(int(s))) < 1e-12
g9["ema_decay"] = np.array(dec)
g9["lr_its"] = np.array([0, 2500, 5000, 5001, 400000, 799999])
g9["lr_ratio"] = np.array([1 / 5000, 2501 / 5000, 5001 / 5000, (1 - 1 / 800000) ** 0.96, (1 - 395000 / 800000) ** 0.96,
                           max((1 - 794999 / 800000) ** 0.96, 5e-6 / 1e-4)])
for i, r in zip(g9["lr_its"], g9["lr_ratio"]):
    assert abs(ddm_ref.lr_lambda(int(i), 1e-4, 5e-6, 800000) - r) < 1e-12
np.savez_compressed(os.path.join(OUT, "g9_schedules.npz"), **g9)
report["cases"].append(dict(case="G9/ema_decay+lr_lambda", ok=True))

report["all_ok"] = all(c["ok"] for c in report["cases"])
with open(os.path.join(OUT, "oracle_vs_reference_report.json"), "w") as f:
    json.dump(report, f, indent=1)
print("ALL OK", len(report["cases"]), "cases")
