"""GPU: the HIP AugmentPipe (adm_amd/ddm/augment.py -> adm_augment_geometric) against the reference AugmentPipe's outputs
on recorded draws (tests/golden/g12_augment.npz) and against the CPU oracle on fresh draws; plus its wiring into
DDPM.p_losses (use_augment: True).  rtol 1e-3 / atol 1e-4."""
import os

import numpy as np
import pytest
import torch

from oracle import augment_ref as A, fill

from parity import close  # noqa: E402  (tests/parity.py: the north_star tolerance, elementwise)

pytestmark = pytest.mark.gpu
KW = dict(xflip=1e8, yflip=1, scale=1, rotate_frac=1, aniso=1, translate_frac=1)


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import hip
    hip.lib()
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag,p,N,H,W,seed,force", [("p012", 0.12, 16, 32, 32, 11, 0.0), ("p015", 0.15, 16, 32, 32, 12, 0.0),
                                                   ("forced", 0.12, 8, 32, 32, 13, 0.9), ("forced64", 0.15, 4, 64, 64, 14, 0.9),
                                                   ("identity", 0.12, 4, 32, 32, 15, -2.0)])
def test_augment_vs_reference_golden(gpu, golden_dir, tag, p, N, H, W, seed, force):
    from adm_amd.ddm.augment import AugmentPipe
    g = np.load(os.path.join(golden_dir, "g12_augment.npz"))
    pipe = AugmentPipe(p=p, **KW)
    assert pipe.label_dim == 9
    x = fill.hash_tensor((N, 3, H, W), f"aug.{tag}.x", 1.0)
    y, lab = pipe(x.to(gpu), draws=A.make_draws(N, seed, force))
    assert y.shape == x.shape and lab.shape == (N, 9)
    close(lab, g[f"{tag}.labels"], rtol=1e-5, atol=1e-6)
    close(y, g[f"{tag}.images"])


def test_augment_vs_oracle_fresh_draws_bs128(gpu):
    """The benchmark's batch shape, every transform firing on most images; also a 1-channel case."""
    from adm_amd.ddm.augment import AugmentPipe
    for N, C, seed in ((128, 3, 21), (5, 1, 22)):
        x = fill.hash_tensor((N, C, 32, 32), f"aug.fresh{seed}", 1.0)
        d = A.make_draws(N, seed, 0.8)
        y_o, lab_o = A.augment(x, d, 0.15)
        y, lab = AugmentPipe(p=0.15, **KW)(x.to(gpu), draws=d)
        close(lab, lab_o, rtol=1e-5, atol=1e-6)
        close(y, y_o)


def test_augment_own_rng_statistics(gpu):
    """Without injected draws: the gates fire with probability ~p (x-flip: 1/2) and labels are 0 where they do not."""
    from adm_amd.ddm.augment import AugmentPipe
    torch.manual_seed(5)
    x = torch.rand(4096, 3, 8, 8, device=gpu) * 2 - 1
    y, lab = AugmentPipe(p=0.12, **KW)(x)
    assert y.shape == x.shape and torch.isfinite(y).all()
    frac = (lab != 0).float().mean(0).cpu()
    assert abs(float(frac[0]) - 0.5) < 0.05                          # x-flip: fair coin
    assert abs(float(frac[1]) - 0.06) < 0.02                         # y-flip: p/2
    for col in (2, 4, 7, 8):                                         # scale, rotation (sin), translation: p
        assert abs(float(frac[col]) - 0.12) < 0.03, (col, float(frac[col]))


def test_ddpm_use_augment_feeds_images_and_labels(gpu):
    """p_losses with use_augment: the UNet must see the augmented x_start and the 9 labels (ddm_const.py:314-316)."""
    from adm_amd.ddm.ddm_const import DDPM
    from adm_amd.unet.uncond_unet import EDMPrecond
    from oracle import ddm_ref, unet_ref
    cfg = unet_ref.default_cfg(variant="uncond_unet", model_channels=64, num_blocks=1, dropout=0.0)
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions",
                              "dropout", "augment_dim")}
    unet = EDMPrecond(img_resolution=32, img_channels=3, model_type="DhariwalUNet", **kw)
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    unet.load_state_dict(sd, strict=True)
    mcfg = dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True, use_augment=True)
    dpm = DDPM(model=unet, image_size=[32, 32], sampling_timesteps=10, perceptual_weight=0.0, cfg=mcfg).to(gpu)
    assert dpm.augment.p == 0.15
    x0 = fill.hash_tensor((4, 3, 32, 32), "x0", 1.0)
    noise = fill.hash_tensor((4, 3, 32, 32), "noise", 1.7)
    t = torch.tensor([0.23, 0.81, 0.5, 0.05])
    d = A.make_draws(4, 31, 0.8)
    loss, _ = dpm.training_step({"image": x0.to(gpu)}, t=t.to(gpu), noise=noise.to(gpu), augment_draws=d)
    xa, lab = A.augment(x0, d, 0.15)
    assert float(lab.abs().max()) > 0
    with torch.no_grad():
        mf = lambda x, tt, **k: unet_ref.edm_precond(sd, cfg, x, tt, **k)
        loss_o, _, _ = ddm_ref.p_losses("const", mf, xa, t, noise, 1e-4, True, augment_labels=lab)
    assert abs(float(loss.detach()) - float(loss_o)) <= 1e-3 * abs(float(loss_o))
