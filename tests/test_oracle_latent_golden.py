"""CPU: the oracle restatement of the KL autoencoder + LatentDiffusion (oracle/ae_ref.py) against the vectors the imported
reference produced (tools/make_golden_latent.py -> tests/golden/g10_autoencoder.npz, g11_latent.npz), plus host-side
checks of the HIP drop-in's module tree (state-dict names, alias dotted paths)."""
import os

import numpy as np
import torch

from oracle import ae_ref, fill, unet_ref


def close(got, want, tol=2e-5):
    got, want = got.detach().double(), torch.as_tensor(np.asarray(want)).double()
    err = float((got - want).abs().max() / (want.abs().max() + 1e-30))
    assert err <= tol, err


def test_autoencoder_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_autoencoder.npz"))
    for tag, cfg, B in (("small", ae_ref.ae_cfg(ch=32, resolution=(32, 32)), 2), ("rect", ae_ref.ae_cfg(ch=32, resolution=(64, 32)), 1)):
        sd = fill.filled_state_dict(ae_ref.param_shapes(cfg))
        H, W = cfg["resolution"]
        x = fill.hash_tensor((B, 3, H, W), f"ae.{tag}.x", 1.0)
        eps = fill.hash_tensor((B, 3, H // 4, W // 4), f"ae.{tag}.eps", 1.7)
        with torch.no_grad():
            mom = ae_ref.encode_moments(sd, cfg, x)
            close(mom, g[f"{tag}.moments"])
            close(ae_ref.posterior_sample(mom, eps), g[f"{tag}.z"])
            close(ae_ref.decode(sd, cfg, torch.from_numpy(g[f"{tag}.z"])), g[f"{tag}.rec"])


def _latent_setup():
    cfg_ae = ae_ref.ae_cfg(ch=32, resolution=(64, 64))
    sd_ae = fill.filled_state_dict(ae_ref.param_shapes(cfg_ae))
    cfg_u = unet_ref.default_cfg(variant="uncond_unet_sd_2", model_channels=64, num_blocks=1, dropout=0.0, img_resolution=16,
                                 attn_resolutions=[8])
    sd_u = fill.filled_state_dict(unet_ref.param_shapes(cfg_u))
    return cfg_ae, sd_ae, cfg_u, sd_u


def test_latent_diffusion_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g11_latent.npz"))
    cfg_ae, sd_ae, cfg_u, sd_u = _latent_setup()
    x = fill.hash_tensor((2, 3, 64, 64), "ldm.x", 1.0)
    eps_enc = fill.hash_tensor((2, 3, 16, 16), "ldm.eps_enc", 1.7)
    noise = fill.hash_tensor((2, 3, 16, 16), "ldm.noise", 1.7)
    t = torch.tensor([0.23, 0.81])
    with torch.no_grad():
        z = ae_ref.posterior_sample(ae_ref.encode_moments(sd_ae, cfg_ae, x), eps_enc)
    close(z, g["z"])
    sf = float(ae_ref.std_scale_factor(z))
    assert abs(sf - float(g["scale_factor"])) <= 1e-6 * sf
    sdo = {k: v.clone().requires_grad_("resample" not in k) for k, v in sd_u.items()}
    mf = lambda xx, tt, **k: unet_ref.edm_precond(sdo, cfg_u, xx, tt, **k)
    loss, log, _ = ae_ref.latent_p_losses(mf, sf * z, t, noise, 1e-3, True)
    loss.backward()
    close(loss, g["loss"]); close(log["train/loss_simple"], g["loss_simple"]); close(log["train/loss_vlb"], g["loss_vlb"])
    close(log["train/loss"], g["log_loss"])
    gn = torch.sqrt(sum(v.grad.double().pow(2).sum() for v in sdo.values() if v.grad is not None))
    close(gn, g["grad_norm"], 1e-4)
    close(sdo["model.map_layer1.bias"].grad, g["grad.map_layer1.bias"], 1e-4)
    # the [B,1] x [B] broadcast of the reference (ddm_const_2.py:566-568): loss_vlb sums a B x B outer product
    xT = fill.hash_tensor((2, 3, 16, 16), "ldm.xT", 1.7, torch.float64)
    mfn = lambda xx, tt: unet_ref.edm_precond(sd_u, cfg_u, xx, tt)
    with torch.no_grad():
        zd = ae_ref.latent_sample_fn_d(mfn, xT, 10, 1.0)
        close(zd, g["sample_d.z"], 1e-4)
        close(ae_ref.latent_sample(sd_ae, cfg_ae, zd, sf), g["sample_d.img"], 1e-4)
        draws = [fill.hash_tensor((2, 3, 16, 16), f"ldm.s{k}", 1.7) for k in range(12)]
        zs = ae_ref.latent_sample_fn_s(mfn, draws[0], draws[1:], 10, 1e-3, denoise=True)
        close(zs, g["sample_s.z"], 1e-4)
        close(ae_ref.latent_sample(sd_ae, cfg_ae, zs, sf), g["sample_s.img"], 1e-4)


def test_hip_autoencoder_module_tree_matches_reference_names():
    """The drop-in's state_dict must have the reference's names / shapes / order (minus loss.*), and resolve through the
    reference's dotted paths."""
    import importlib
    ED = importlib.import_module("ddm.encoder_decoder")
    D2 = importlib.import_module("ddm.ddm_const_2")
    assert hasattr(D2, "LatentDiffusion") and hasattr(D2, "DDPM")
    for ch, res in ((32, (32, 32)), (128, (256, 256))):
        cfg = ae_ref.ae_cfg(ch=ch, resolution=res)
        dd = dict(double_z=True, z_channels=3, resolution=list(res), in_channels=3, out_ch=3, ch=ch, ch_mult=[1, 2, 4],
                  num_res_blocks=2, attn_resolutions=[], dropout=0.0)
        ae = ED.AutoencoderKL(dd, dict(disc_start=20001, kl_weight=1e-6, disc_weight=0.5), 3)
        sd, shapes = ae.state_dict(), ae_ref.param_shapes(cfg)
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
        assert ae.down_ratio == 4


def test_hip_autoencoder_refuses_cpu_tensors():
    import importlib
    import pytest
    ED = importlib.import_module("ddm.encoder_decoder")
    dd = dict(double_z=True, z_channels=3, resolution=[32, 32], in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    ae = ED.AutoencoderKL(dd, {}, 3)
    with pytest.raises(RuntimeError):
        ae.encode(torch.zeros(1, 3, 32, 32))


def test_augment_oracle_vs_golden(golden_dir):
    """oracle/augment_ref.py against the reference AugmentPipe's outputs on recorded draws (g12)."""
    from oracle import augment_ref as A
    g = np.load(os.path.join(golden_dir, "g12_augment.npz"))
    for tag, p, N, H, W, seed, force in (("p012", 0.12, 16, 32, 32, 11, 0.0), ("forced", 0.12, 8, 32, 32, 13, 0.9),
                                          ("forced64", 0.15, 4, 64, 64, 14, 0.9), ("identity", 0.12, 4, 32, 32, 15, -2.0)):
        x = fill.hash_tensor((N, 3, H, W), f"aug.{tag}.x", 1.0)
        y, lab = A.augment(x, A.make_draws(N, seed, force), p)
        close(lab, g[f"{tag}.labels"], 1e-6)
        close(y, g[f"{tag}.images"])
    # with no transform firing the pipe is the sym6 up/down round trip: near-identity, not exact
    x = fill.hash_tensor((4, 3, 32, 32), "aug.identity.x", 1.0)
    lab = torch.from_numpy(g["identity.labels"])
    assert float(lab[:, 1:].abs().max()) == 0                      # only the x-flip (gate multiplier 1e8: a fair coin) fires
    xf = torch.where(lab[:, 0].reshape(-1, 1, 1, 1) == 1, x.flip(3), x)
    assert float((torch.from_numpy(g["identity.images"]) - xf).abs().max()) < 0.15
