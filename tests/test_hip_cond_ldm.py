"""GPU: the conditional latent wrapper of BASELINE configs[4] (ddm.ddm_const.LatentDiffusion + unet.cond_unet_sd.Unet + frozen KL-f4
autoencoder): training step (const schedule, use_l1, condition features) and the 5-step deterministic sampler against the CPU
oracle; the sliding-window stitching against the oracle's restatement of /root/reference/sample_cond_ldm.py:281-330; the
sample_cond_ldm.py CLI end to end on a reduced model.  CPU (not gpu): window geometry + stitching with a closed-form sampler."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from oracle import ae_ref, fill
from oracle import cond_unet_ref as R
from parity import close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stitching_matches_the_oracle_on_cpu():
    """sample_cond_ldm.slide_sample_sr (batched windows) == the oracle's one-window-at-a-time restatement, for a sampler whose
    output depends on the window content AND position-independent noise-free arithmetic."""
    from sample_cond_ldm import slide_sample_sr, slide_windows
    assert slide_windows(200, 136, (128, 128), (64, 64)) == R.slide_windows(200, 136, (128, 128), (64, 64))
    cond = fill.hash_tensor((2, 3, 40, 28), "sw.c", 1.0)
    fn = lambda c: torch.nn.functional.interpolate(c, scale_factor=4, mode="bilinear", align_corners=False) * c.mean((1, 2, 3), keepdim=True)
    want = R.slide_sample_sr(fn, cond, (160, 112), (16, 16), (8, 8), ori_size=(150, 100))
    for wb in (0, 1, 3):
        got = slide_sample_sr(fn, cond, (160, 112), (16, 16), (8, 8), ori_size=(150, 100), window_batch=wb)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    flip = slide_sample_sr(fn, cond, (160, 112), (16, 16), (8, 8), window_batch=0, flip_test=True)
    assert flip.shape == (2, 3, 160, 112)


def build_ldm(gpu, two_decoders=False, steps=5):
    ED = importlib.import_module("ddm.encoder_decoder")
    D = importlib.import_module("ddm.ddm_const")
    U = importlib.import_module("unet.cond_unet" if two_decoders else "unet.cond_unet_sd")
    dd = dict(double_z=True, z_channels=3, resolution=[128, 128], in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 4], num_res_blocks=2,
              attn_resolutions=[], dropout=0.0)
    ae = ED.AutoencoderKL(dd, dict(disc_start=50001, kl_weight=1e-6, disc_weight=0.5), 3)
    ae.load_state_dict(fill.filled_state_dict(ae_ref.param_shapes(ae_ref.ae_cfg(ch=32, resolution=(128, 128)))), strict=True)
    cfg = R.default_cfg(dim=32, two_decoders=two_decoders)
    unet = U.Unet(dim=32, dim_mults=cfg["dim_mults"], cond_dim=32, cond_dim_mults=(), channels=3, cond_in_dim=3,
                  window_sizes1=cfg["window_sizes1"], window_sizes2=cfg["window_sizes2"], fourier_scale=16, cfg={"cond_net": "swin"})
    sd = R.filled_state_dict(cfg)
    unet.load_state_dict(sd, strict=True)
    for m in unet.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    mcfg = dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True, use_augment=False, ldm=True)
    ldm = D.LatentDiffusion(auto_encoder=ae, scale_factor=0.195, scale_by_std=True, default_scale=True, model=unet,
                            image_size=[128, 128], sampling_timesteps=steps, loss_type="l2", start_dist="normal",
                            perceptual_weight=0.0, use_l1=True, cfg=dict(mcfg))
    return ldm.to(gpu), cfg, sd


@pytest.mark.gpu
def test_conditional_latent_training_step_and_sampler_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    gpu = torch.device("cuda:0")
    ldm, cfg, sd = build_ldm(gpu)
    ldm.eval()              # BatchNorm on running statistics: the oracle call below is made with training=False
    assert ldm.SCHEDULE == "const" and ldm.use_l1 and abs(float(ldm.scale_factor) - 0.195) < 1e-7
    B = 2
    z = fill.hash_tensor((B, 3, 32, 32), "cl.z", 1.0)
    noise = fill.hash_tensor((B, 3, 32, 32), "cl.noise", 1.7)
    t = torch.tensor([0.23, 0.81])
    hm = R.cond_features(B, 32, 32)
    loss, log = ldm.p_losses(z.to(gpu), t.to(gpu), [h.to(gpu) for h in hm], noise=noise.to(gpu))
    loss.backward()
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "time_mlp.0.W" else v.clone())
           for k, v in sd.items()}
    mf = lambda x, tt: R.unet_forward(sdo, cfg, x, tt, hm)
    loss_o, log_o = R.latent_p_losses_const(mf, z, t, noise, eps=1e-4, weighting_loss=True, use_l1=True)
    loss_o.backward()
    assert abs(float(loss) - float(loss_o)) <= 1e-3 * abs(float(loss_o)), (float(loss), float(loss_o))
    for k in ("train/loss_simple", "train/loss_vlb"):
        assert abs(float(log[k]) - float(log_o[k])) <= 1e-3 * abs(float(log_o[k])), k
    gmax = max(float(v.grad.double().norm()) for v in sdo.values() if v.requires_grad and v.grad is not None)
    bad = []
    for name, p in ldm.model.named_parameters():
        if not p.requires_grad:
            continue
        want = sdo[name].grad
        err = float((p.grad.cpu().double() - want.double()).norm() / (want.double().norm() + 1e-4 * gmax))
        if err > 2e-3:
            bad.append((name, err))
    assert not bad, bad[:10]
    assert all(p.grad is None for p in ldm.first_stage_model.parameters())
    # 5-step deterministic latent sampler with the condition features (fp64 state)
    xT = fill.hash_tensor((B, 3, 32, 32), "cl.xT", 1.7, torch.float64)
    zs = ldm.sample_fn_d((B, 3, 32, 32), unnormalize=False, x_T=xT.to(gpu), cond=[h.to(gpu) for h in hm])
    assert zs.dtype == torch.float64
    with torch.no_grad():
        sd0 = {k: v.detach() for k, v in sdo.items()}
        want = R.latent_sample_fn_d_const(lambda x, tt: R.unet_forward(sd0, cfg, x, tt, hm), xT, 5, 0.01, 1.0)
    close(zs, want)
    img = ldm.sample(cond=[h.to(gpu) for h in hm], x_T=xT.to(gpu))
    assert img.shape == (B, 3, 128, 128) and float(img.min()) >= 0 and float(img.max()) <= 1
    with torch.no_grad():
        dec = ae_ref.decode(fill.filled_state_dict(ae_ref.param_shapes(ae_ref.ae_cfg(ch=32, resolution=(128, 128)))),
                            ae_ref.ae_cfg(ch=32, resolution=(128, 128)), (want / 0.195).float())
    close(img, ((dec + 1) * 0.5).clamp(0, 1))


@pytest.mark.gpu
def test_sample_cond_ldm_cli_end_to_end(tmp_path):
    """YAML -> construct by dotted name -> sliding-window SR of two synthetic 96x80 images (24x20 conditions, 16x16 windows,
    stride 8, synthetic condition encoder, random init) -> PNGs of the ORIGINAL size."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cfg = yaml.load(open(os.path.join(ROOT, "configs/super-resolution/div2k_cond_ddm_const_ldm.yaml")), Loader=yaml.SafeLoader)
    cfg["model"].update(image_size=[64, 64], sampling_timesteps=2)
    cfg["model"]["first_stage"]["ddconfig"].update(ch=32, resolution=[64, 64])
    cfg["model"]["unet"].update(dim=32, class_name="unet.cond_unet_sd.Unet", window_sizes1=[[2, 2], [1, 1], [1, 1], [1, 1]],
                                window_sizes2=[[4, 4], [2, 2], [1, 1], [1, 1]])
    cfg["data"].update(image_size=[94, 78])
    out = str(tmp_path / "png")
    cfg["sampler"].update(sample_num=2, crop_size=[16, 16], stride=[8, 8], save_folder=out, cond_encoder="synthetic", window_batch=4)
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sample_cond_ldm.py"), "--cfg", path, "--random-init"], capture_output=True,
                       text=True, env=dict(os.environ, PYTHONPATH=ROOT), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "PSNR:" in r.stdout
    from PIL import Image
    names = sorted(os.listdir(out))
    assert names == [f"{i: 010d}.png" for i in range(2)], names
    assert Image.open(os.path.join(out, names[0])).size == (78, 94)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sample_cond_ldm.py"), "--cfg", path], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=ROOT), timeout=900)
    assert r.returncode != 0 and "does not exist" in (r.stdout + r.stderr)      # a missing checkpoint is an error, not a silent random init
