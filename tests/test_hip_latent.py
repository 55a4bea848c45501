"""GPU parity tests for the latent path: strided conv / row softmax / posterior kernels against plain PyTorch fp32, the
HIP KL autoencoder against the reference's golden vectors (g10) and the CPU oracle, and LatentDiffusion
(training_step, both samplers -> decoded images) against g11.  rtol 1e-3 / atol 1e-4 fp32 (north_star)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ae_ref, fill, unet_ref

from parity import close  # noqa: E402  (tests/parity.py: the north_star tolerance, elementwise)

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-3, 1e-4


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def nhwc(x, cpad):
    B, C, H, W = x.shape
    y = torch.zeros(B, H, W, cpad)
    y[..., :C] = x.permute(0, 2, 3, 1)
    return y


@pytest.mark.parametrize("B,H,W,ci,co", [(2, 16, 16, 32, 32), (1, 32, 16, 64, 96), (3, 8, 8, 128, 64), (1, 6, 10, 32, 32)])
def test_strided_conv_vs_torch(gpu, B, H, W, ci, co):
    """pad (0,1,0,1) + 3x3 stride 2 (encoder_decoder.py:78-96), odd and even sizes."""
    from adm_amd import ops
    x = fill.hash_tensor((B, ci, H, W), "sx", 1.0)
    w = fill.hash_tensor((co, ci, 3, 3), "sw", (1.0 / (9 * ci)) ** 0.5)
    b = fill.hash_tensor((co,), "sb", 0.1)
    want = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    with torch.no_grad():
        y = ops.conv2d_strided(nhwc(x, ci).to(gpu), w.to(gpu), b.to(gpu), stride=2, pad_lo=0, pad_hi=1)
    assert y.shape == (B, want.shape[2], want.shape[3], co)
    close(y.permute(0, 3, 1, 2), want)
    # symmetric padding 1, stride 2 as a second geometry
    want = F.conv2d(x, w, b, stride=2, padding=1)
    with torch.no_grad():
        y = ops.conv2d_strided(nhwc(x, ci).to(gpu), w.to(gpu), b.to(gpu), stride=2, pad_lo=1, pad_hi=1)
    close(y.permute(0, 3, 1, 2), want)


def test_strided_conv_is_forward_only(gpu):
    from adm_amd import ops
    w = torch.nn.Parameter(torch.zeros(32, 32, 3, 3, device=gpu))
    with pytest.raises(RuntimeError):
        ops.conv2d_strided(torch.zeros(1, 8, 8, 32, device=gpu), w, None)


@pytest.mark.parametrize("rows,cols", [(64, 64), (300, 1024), (5, 4096), (17, 8192), (9, 36), (7, 16384), (3, 8196)])   # > 8192: the three-pass kernel
def test_softmax_rows_vs_torch(gpu, rows, cols):
    from adm_amd import ops
    s = fill.hash_tensor((rows, cols), "sm", 6.0)
    got = ops.softmax_rows_(s.clone().to(gpu), 0.37)
    close(got, torch.softmax(0.37 * s.double(), dim=1), rtol=1e-5, atol=1e-6)
    assert float((got.sum(1) - 1).abs().max()) < 1e-5


def test_matmul_nt_and_posterior(gpu):
    from adm_amd import ops
    a, b = fill.hash_tensor((96, 64), "mma", 1.0), fill.hash_tensor((160, 64), "mmb", 1.0)
    bias = fill.hash_tensor((160,), "mmbias", 1.0)
    with torch.no_grad():
        y = ops.matmul_nt(a.to(gpu), b.to(gpu), bias.to(gpu))
    close(y, a.double() @ b.double().T + bias.double(), rtol=1e-5, atol=1e-6)
    mom = fill.hash_tensor((2, 4, 4, 32), "mom", 2.0)
    mom[..., 3:6] *= 20                                   # exercise the logvar clamp
    eps = fill.hash_tensor((2, 4, 4, 3), "meps", 1.7)
    z = ops.posterior_sample(mom.to(gpu), 3, eps.to(gpu), 1.0)
    want = mom[..., :3] + torch.exp(0.5 * mom[..., 3:6].clamp(-30, 20)) * eps
    close(z, want, rtol=1e-5, atol=1e-6)
    close(ops.posterior_sample(mom.to(gpu), 3, None), mom[..., :3], rtol=0, atol=0)


def build_ae(gpu, ch, res):
    ED = importlib.import_module("ddm.encoder_decoder")            # the reference's dotted path (alias package)
    cfg = ae_ref.ae_cfg(ch=ch, resolution=res)
    dd = dict(double_z=True, z_channels=3, resolution=list(res), in_channels=3, out_ch=3, ch=ch, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    ae = ED.AutoencoderKL(dd, dict(disc_start=20001, kl_weight=1e-6, disc_weight=0.5), 3)
    sd = fill.filled_state_dict(ae_ref.param_shapes(cfg))
    ae.load_state_dict(sd, strict=True)
    return ae.to(gpu).eval(), cfg, sd


@pytest.mark.parametrize("tag,ch,res,B", [("small", 32, (32, 32), 2), ("rect", 32, (64, 32), 1), ("klf4", 128, (32, 32), 1)])
def test_autoencoder_vs_golden(gpu, golden_dir, tag, ch, res, B):
    g = np.load(os.path.join(golden_dir, "g10_autoencoder.npz"))
    ae, cfg, _ = build_ae(gpu, ch, res)
    x = fill.hash_tensor((B, 3, *res), f"ae.{tag}.x", 1.0)
    eps = fill.hash_tensor((B, 3, res[0] // 4, res[1] // 4), f"ae.{tag}.eps", 1.7)
    post = ae.encode(x.to(gpu))
    close(post.parameters, g[f"{tag}.moments"])
    z = post.sample(eps.to(gpu))
    assert z.shape == (B, 3, res[0] // 4, res[1] // 4)
    close(z, g[f"{tag}.z"])
    close(post.mode(), g[f"{tag}.moments"][:, :3], scale=float(np.abs(g[f"{tag}.moments"]).max()))
    rec = ae.decode(torch.from_numpy(g[f"{tag}.z"]).to(gpu))
    assert rec.shape == (B, 3, *res)
    close(rec, g[f"{tag}.rec"])


def test_autoencoder_chunked_batch_equals_whole(gpu):
    """encode/decode split big batches into passes; the split must not change results."""
    from adm_amd.ddm import encoder_decoder as E
    ae, cfg, _ = build_ae(gpu, 32, (32, 32))
    x = fill.hash_tensor((5, 3, 32, 32), "chunk.x", 1.0).to(gpu)
    whole = ae.encode(x).parameters
    old = E._CHUNK_BYTES
    try:
        E._CHUNK_BYTES = 2 * 32 * 32 * 128 * 4          # two images per pass
        parts = ae.encode(x).parameters
        rec_parts = ae.decode(whole[:, :3])
    finally:
        E._CHUNK_BYTES = old
    assert torch.equal(whole, parts)
    assert torch.equal(rec_parts, ae.decode(whole[:, :3]))


def build_ldm(gpu):
    D2 = importlib.import_module("ddm.ddm_const_2")
    U = importlib.import_module("unet.uncond_unet_sd_2")
    ae, cfg_ae, sd_ae = build_ae(gpu, 32, (64, 64))
    cfg_u = unet_ref.default_cfg(variant="uncond_unet_sd_2", model_channels=64, num_blocks=1, dropout=0.0, img_resolution=16,
                                 attn_resolutions=[8])
    kw = {k: cfg_u[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions",
                                "dropout", "augment_dim")}
    unet = U.EDMPrecond(img_resolution=16, img_channels=3, model_type="DhariwalUNet", **kw)
    sd_u = fill.filled_state_dict(unet_ref.param_shapes(cfg_u))
    unet.load_state_dict(sd_u, strict=True)
    model_cfg = dict(eps=1e-3, sigma_max=1, sigma_min=0.001, weighting_loss=True, use_augment=False, use_disloss=False)
    ldm = D2.LatentDiffusion(auto_encoder=ae, scale_factor=1.0, scale_by_std=True, default_scale=False, model=unet,
                             image_size=[64, 64], sampling_timesteps=10, loss_type="l2", start_dist="normal",
                             perceptual_weight=0.0, use_l1=False, cfg=dict(model_cfg))
    return ldm.to(gpu), cfg_ae, sd_ae, cfg_u, sd_u


def test_latent_training_step_vs_golden(gpu, golden_dir):
    g = np.load(os.path.join(golden_dir, "g11_latent.npz"))
    ldm, *_ = build_ldm(gpu)
    ldm.train()
    assert all(not p.requires_grad for p in ldm.first_stage_model.parameters())
    x = fill.hash_tensor((2, 3, 64, 64), "ldm.x", 1.0).to(gpu)
    eps_enc = fill.hash_tensor((2, 3, 16, 16), "ldm.eps_enc", 1.7).to(gpu)
    noise = fill.hash_tensor((2, 3, 16, 16), "ldm.noise", 1.7).to(gpu)
    t = torch.tensor([0.23, 0.81], device=gpu)
    ldm.on_train_batch_start({"image": x}, eps=eps_enc)
    sf = float(g["scale_factor"])
    assert abs(float(ldm.scale_factor) - sf) <= 1e-4 * sf
    z, _, _ = ldm.get_input({"image": x}, eps=eps_enc)
    close(z, g["z"])
    loss, log = ldm.training_step({"image": x}, eps=eps_enc, t=t, noise=noise)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
    assert abs(float(log["train/loss_simple"]) - float(g["loss_simple"])) <= 1e-3 * float(g["loss_simple"])
    assert abs(float(log["train/loss_vlb"]) - float(g["loss_vlb"])) <= 1e-3 * float(g["loss_vlb"])
    assert abs(float(log["train/loss"]) - float(g["log_loss"])) <= 1e-3 * float(g["log_loss"])
    gn = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in ldm.model.parameters() if p.grad is not None)))
    assert abs(gn - float(g["grad_norm"])) <= 2e-3 * float(g["grad_norm"])
    got = dict(ldm.model.named_parameters())["model.map_layer1.bias"].grad
    close(got, g["grad.map_layer1.bias"], rtol=2e-3, atol=2e-4)
    assert all(p.grad is None for p in ldm.first_stage_model.parameters())


def test_latent_sample_vs_golden(gpu, golden_dir):
    g = np.load(os.path.join(golden_dir, "g11_latent.npz"))
    ldm, *_ = build_ldm(gpu)
    ldm.eval()
    del ldm.scale_factor
    ldm.register_buffer("scale_factor", torch.tensor(float(g["scale_factor"]), device=gpu))
    xT = fill.hash_tensor((2, 3, 16, 16), "ldm.xT", 1.7, torch.float64)
    ldm.cfg["sample_type"] = "deterministic"
    z = ldm.sample_fn_d((2, 3, 16, 16), unnormalize=False, x_T=xT.to(gpu))
    close(z, g["sample_d.z"])
    img = ldm.sample(batch_size=2, x_T=xT.to(gpu))
    assert img.shape == (2, 3, 64, 64) and float(img.min()) >= 0 and float(img.max()) <= 1
    close(img, g["sample_d.img"])
    draws = [fill.hash_tensor((2, 3, 16, 16), f"ldm.s{k}", 1.7) for k in range(12)]
    ldm.cfg["sample_type"] = "stochastic"
    zs = ldm.sample_fn_s((2, 3, 16, 16), unnormalize=False, denoise=True, x_T=draws[0].to(gpu), epsilons=draws[1:])
    close(zs, g["sample_s.z"])
    img = ldm.sample(batch_size=2, x_T=draws[0].to(gpu), epsilons=draws[1:])
    close(img, g["sample_s.img"])
