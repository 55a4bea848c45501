#!/usr/bin/env python3
"""Golden vectors for the latent path (KL autoencoder + ddm_const_2.LatentDiffusion).
RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference); same import shims as tools/make_golden.py.

The reference's AutoencoderKL cannot be constructed offline: its __init__ builds LPIPSWithDiscriminator, whose LPIPS
backbone calls torchvision.models.vgg16(pretrained=True) (ddm/encoder_decoder.py:908, taming/modules/losses/lpips.py:16).
Its arithmetic is Encoder / Decoder / two 1x1 convs (encoder_decoder.py:937-946), so the script builds the reference
Encoder and Decoder classes directly, composes encode/decode exactly as those five lines do in a 12-line holder
module, and hands that holder to the reference LatentDiffusion as `auto_encoder`.

Writes tests/golden/g10_autoencoder.npz, g11_latent.npz and oracle_vs_reference_report_latent.json.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
adm = types.ModuleType("ADM"); adm.__path__ = [REF]; sys.modules["ADM"] = adm
tv = types.ModuleType("torchvision"); tv.models = types.ModuleType("torchvision.models")
tv.transforms = types.ModuleType("torchvision.transforms")
sys.modules["torchvision"] = tv; sys.modules["torchvision.models"] = tv.models
sys.modules["torchvision.transforms"] = tv.transforms

import importlib  # noqa: E402

from oracle import ae_ref, fill, unet_ref  # noqa: E402

import ddm.ddm_const_2 as D2  # noqa: E402
import ddm.encoder_decoder as ED  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
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


class RefFirstStage(torch.nn.Module):
    """encode()/decode()/down_ratio of the reference AutoencoderKL (encoder_decoder.py:894-946) around the REAL
    reference Encoder / Decoder, without the LPIPS/discriminator training loss."""

    def __init__(self, ddconfig, embed_dim):
        super().__init__()
        self.encoder = ED.Encoder(**ddconfig)
        self.decoder = ED.Decoder(**ddconfig)
        self.down_ratio = 2 ** (len(ddconfig["ch_mult"]) - 1)
        self.quant_conv = torch.nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = torch.nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)

    def encode(self, x):
        return ED.DiagonalGaussianDistribution(self.quant_conv(self.encoder(x)))

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def build_first_stage(cfg):
    dd = dict(double_z=True, z_channels=cfg["z_channels"], resolution=list(cfg["resolution"]), in_channels=cfg["in_channels"],
              out_ch=cfg["out_ch"], ch=cfg["ch"], ch_mult=list(cfg["ch_mult"]), num_res_blocks=cfg["num_res_blocks"],
              attn_resolutions=[], dropout=0.0)
    fs = RefFirstStage(dd, cfg["embed_dim"]).eval()
    shapes = ae_ref.param_shapes(cfg)
    ref_sd = fs.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), [k for k in ref_sd if k not in shapes][:5]
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sd = fill.filled_state_dict(shapes)
    fs.load_state_dict(sd, strict=True)
    return fs, sd


# ------------------------------------------------------------------------------------------------
# G10: autoencoder, reduced width (ch=32) at 32x32 and 64x48 (non-square), plus the KL-f4 width (ch=128) at 32x32
# ------------------------------------------------------------------------------------------------
g10 = {}
for tag, cfg, B in (("small", ae_ref.ae_cfg(ch=32, resolution=(32, 32)), 2),
                    ("rect", ae_ref.ae_cfg(ch=32, resolution=(64, 32)), 1),
                    ("klf4", ae_ref.ae_cfg(ch=128, resolution=(32, 32)), 1)):
    fs, sd = build_first_stage(cfg)
    H, W = cfg["resolution"]
    x = fill.hash_tensor((B, 3, H, W), f"ae.{tag}.x", 1.0)
    eps = fill.hash_tensor((B, 3, H // 4, W // 4), f"ae.{tag}.eps", 1.7)
    with torch.no_grad():
        post = fs.encode(x)
        mom_ref = post.parameters
        _randn = torch.randn
        torch.randn = lambda *a, **k: eps.clone()
        z_ref = post.sample()
        torch.randn = _randn
        rec_ref = fs.decode(z_ref)
        mom_o = ae_ref.encode_moments(sd, cfg, x)
        z_o = ae_ref.posterior_sample(mom_o, eps)
        rec_o = ae_ref.decode(sd, cfg, z_ref)
    check(f"G10/{tag}/moments", mom_o, mom_ref)
    check(f"G10/{tag}/posterior_sample", z_o, z_ref)
    check(f"G10/{tag}/mode", ae_ref.posterior_sample(mom_o, None), post.mode())
    check(f"G10/{tag}/decode", rec_o, rec_ref)
    g10[f"{tag}.moments"] = mom_ref.numpy(); g10[f"{tag}.z"] = z_ref.numpy(); g10[f"{tag}.rec"] = rec_ref.numpy()
np.savez_compressed(os.path.join(OUT, "g10_autoencoder.npz"), **g10)

# ------------------------------------------------------------------------------------------------
# G11: LatentDiffusion (const_2): get_input/std-rescaling, p_losses (+ gradients), sample (both samplers) -> images
# ------------------------------------------------------------------------------------------------
g11 = {}
cfg_ae = ae_ref.ae_cfg(ch=32, resolution=(64, 64))
fs, sd_ae = build_first_stage(cfg_ae)
cfg_u = unet_ref.default_cfg(variant="uncond_unet_sd_2", model_channels=64, num_blocks=1, dropout=0.0, img_resolution=16,
                             attn_resolutions=[8])
U = importlib.import_module("unet.uncond_unet_sd_2")
kw = {k: cfg_u[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions", "dropout",
                            "augment_dim")}
unet = U.EDMPrecond(img_resolution=16, img_channels=3, model_type="DhariwalUNet", **kw)
shapes_u = unet_ref.param_shapes(cfg_u)
assert list(unet.state_dict().keys()) == list(shapes_u.keys())
sd_u = fill.filled_state_dict(shapes_u)
unet.load_state_dict(sd_u, strict=True)
unet.eval()
model_cfg = dict(eps=1e-3, sigma_max=1, sigma_min=0.001, weighting_loss=True, use_augment=False, use_disloss=False)
ldm = D2.LatentDiffusion(auto_encoder=fs, scale_factor=1.0, scale_by_std=True, default_scale=False, model=unet,
                         image_size=[64, 64], sampling_timesteps=10, loss_type="l2", start_dist="normal",
                         perceptual_weight=0.0, use_l1=False, cfg=dict(model_cfg))
x = fill.hash_tensor((2, 3, 64, 64), "ldm.x", 1.0)
eps_enc = fill.hash_tensor((2, 3, 16, 16), "ldm.eps_enc", 1.7)
noise = fill.hash_tensor((2, 3, 16, 16), "ldm.noise", 1.7)
t = torch.tensor([0.23, 0.81])
_randn, _randn_like = torch.randn, torch.randn_like
torch.randn = lambda *a, **k: eps_enc.clone()
ldm.on_train_batch_start({"image": x})                 # sets scale_factor = 1 / std(z)
sf_ref = float(ldm.scale_factor)
z_ref, _, _ = ldm.get_input({"image": x})
torch.randn = _randn
with torch.no_grad():
    z_o = ae_ref.posterior_sample(ae_ref.encode_moments(sd_ae, cfg_ae, x), eps_enc)
check("G11/get_input/z", z_o, z_ref)
check("G11/scale_factor", ae_ref.std_scale_factor(z_o), torch.tensor(sf_ref))
zs = sf_ref * z_ref
torch.randn_like = lambda *a, **k: noise.clone()
loss_ref, log_ref = ldm.p_losses(zs, t)
torch.randn_like = _randn_like
loss_ref.backward()
sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd_u.items()}
mf = lambda xx, tt, **k: unet_ref.edm_precond(sdo, cfg_u, xx, tt, **k)
loss_o, log_o, _ = ae_ref.latent_p_losses(mf, zs, t, noise, 1e-3, True)
loss_o.backward()
check("G11/p_losses/loss", loss_o, loss_ref)
for k in ("train/loss_simple", "train/loss_vlb", "train/loss"):
    check("G11/p_losses/" + k, log_o[k], log_ref[k])
gn_ref = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in unet.parameters() if p.grad is not None))
gn_o = torch.sqrt(sum(v.grad.double().pow(2).sum() for v in sdo.values() if v.grad is not None))
check("G11/p_losses/grad_norm", gn_o, gn_ref, 1e-4)
assert all(p.grad is None for p in fs.parameters()), "first stage must stay frozen"
g11.update({"scale_factor": np.array(sf_ref), "z": z_ref.numpy(), "loss": loss_ref.detach().numpy(),
            "loss_simple": log_ref["train/loss_simple"].numpy(), "loss_vlb": log_ref["train/loss_vlb"].numpy(),
            "log_loss": log_ref["train/loss"].numpy(), "grad_norm": gn_ref.numpy(),
            "grad.map_layer1.bias": dict(unet.named_parameters())["model.map_layer1.bias"].grad.numpy().copy()})

# sample(): deterministic
xT = fill.hash_tensor((2, 3, 16, 16), "ldm.xT", 1.7, torch.float64)
torch.randn = lambda *a, **k: xT.clone()
ldm.cfg["sample_type"] = "deterministic"
img_ref = ldm.sample(batch_size=2)
torch.randn = _randn
mfn = lambda xx, tt: unet_ref.edm_precond(sd_u, cfg_u, xx, tt)
with torch.no_grad():
    z_d = ae_ref.latent_sample_fn_d(mfn, xT, 10, 1.0)
    img_o = ae_ref.latent_sample(sd_ae, cfg_ae, z_d, sf_ref)
check("G11/sample/deterministic", img_o, img_ref, 1e-4)
g11["sample_d.z"] = z_d.numpy(); g11["sample_d.img"] = img_ref.numpy()

# sample(): stochastic (denoise=True -> 11 model calls); draws injected in call order (x_T, then one per step)
draws = [fill.hash_tensor((2, 3, 16, 16), f"ldm.s{k}", 1.7) for k in range(12)]
it = iter(draws)
torch.randn = lambda *a, **k: next(it).clone()
torch.randn_like = lambda *a, **k: next(it).clone()
ldm.cfg["sample_type"] = "stochastic"
img_ref = ldm.sample(batch_size=2)
torch.randn, torch.randn_like = _randn, _randn_like
with torch.no_grad():
    z_s = ae_ref.latent_sample_fn_s(mfn, draws[0], draws[1:], 10, 1e-3, denoise=True)
    img_o = ae_ref.latent_sample(sd_ae, cfg_ae, z_s, sf_ref)
check("G11/sample/stochastic", img_o, img_ref, 1e-4)
g11["sample_s.z"] = z_s.numpy(); g11["sample_s.img"] = img_ref.numpy()
np.savez_compressed(os.path.join(OUT, "g11_latent.npz"), **g11)

report["all_ok"] = all(c["ok"] for c in report["cases"])
with open(os.path.join(OUT, "oracle_vs_reference_report_latent.json"), "w") as f:
    json.dump(report, f, indent=1)
print("ALL OK", len(report["cases"]), "cases")
