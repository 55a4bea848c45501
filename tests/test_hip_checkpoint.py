"""A checkpoint WRITTEN BY THE REFERENCE's own classes (tools/make_golden_checkpoint.py: reference EDMPrecond + DDPM +
ddm.ema.EMA + AdamW/LambdaLR, saved with the dict layout of /root/reference/train_uncond_dpm.py:207-220) is consumed by
this build's loaders: the sampler path (EMA keys, prefix stripped: /root/reference/sample_uncond.py:131-147) and the
trainer's resume path (model, optimiser moments, EMA shadow, counters).  Expected outputs (g13_checkpoint.npz) are what the
reference model produced from the same file."""
import os

import numpy as np
import pytest
import torch

from oracle import fill
from parity import close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT = os.path.join(ROOT, "tests", "golden", "ref_checkpoint_model-1.pt")
UNET_KW = dict(img_resolution=16, img_channels=3, model_type="DhariwalUNet", model_channels=32, channel_mult=[1],
               channel_mult_emb=2, num_blocks=1, attn_resolutions=[8], dropout=0.0, augment_dim=0)
MODEL_CFG = dict(eps=1e-3, sigma_max=1, sigma_min=0.001, weighting_loss=True, use_augment=False)


def build():
    from ddm.utils import construct_class_by_name
    unet = construct_class_by_name(class_name="unet.uncond_unet_sd_2.EDMPrecond", **UNET_KW)
    return construct_class_by_name(class_name="ddm.ddm_const_2.DDPM", model=unet, image_size=[16, 16], sampling_timesteps=4,
                                   loss_type="l2", start_dist="normal", perceptual_weight=0.0, use_l1=False,
                                   cfg=dict(MODEL_CFG))


def test_reference_checkpoint_keys_match_this_build():
    """CPU: same key set and shapes in 'model' and under both EMA prefixes (ddm/ema.py:66-73), safe loader only."""
    ck = torch.load(CKPT, map_location="cpu", weights_only=True)
    assert set(ck) == {"step", "model", "opt", "lr_scheduler", "ema", "scaler"} and ck["step"] == 6
    mine = build().state_dict()
    assert {k: tuple(v.shape) for k, v in ck["model"].items()} == {k: tuple(v.shape) for k, v in mine.items()}
    for prefix in ("ema_model.", "online_model."):
        got = {k[len(prefix):] for k in ck["ema"] if k.startswith(prefix)}
        assert got == set(mine), prefix
    assert set(ck["ema"]) - {k for k in ck["ema"] if k.startswith(("ema_model.", "online_model."))} == {"initted", "step"}
    assert len(ck["opt"]["state"]) == sum(1 for p in build().parameters() if p.requires_grad)


@pytest.mark.gpu
@pytest.mark.parametrize("use_ema", [True, False])
def test_sampler_loads_reference_checkpoint(use_ema):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sample_uncond import load_weights
    g = np.load(os.path.join(ROOT, "tests", "golden", "g13_checkpoint.npz"))
    dev = torch.device("cuda:0")
    dpm = build().to(dev).eval()
    load_weights(dpm, CKPT, use_ema, dev)
    tag = "ema" if use_ema else "online"
    w0 = dict(dpm.named_parameters())["model.model.enc.16x16_conv.weight"].detach().reshape(-1)[:16]
    close(w0, g[tag + ".w0"], rtol=0, atol=0)
    xT = fill.hash_tensor((2, 3, 16, 16), "ck.xT", 1.7, torch.float64)
    img = dpm.sample(batch_size=2, x_T=xT.to(dev))
    close(img, g[tag + ".img"])
    assert float(np.abs(g["ema.img"] - g["online.img"]).max()) > 1e-6


@pytest.mark.gpu
def test_trainer_resumes_from_reference_checkpoint(tmp_path):
    """Resume: model, AdamW moments (torch's per-parameter state -> the flat buffers), EMA shadow and counters."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import shutil
    from train_uncond_dpm import Cfg, ImageStream, Trainer
    g = np.load(os.path.join(ROOT, "tests", "golden", "g13_checkpoint.npz"))
    dev = torch.device("cuda:0")
    res = tmp_path / "run"
    res.mkdir()
    shutil.copy(CKPT, res / "model-1.pt")
    cfg = Cfg({"trainer": dict(gradient_accumulate_every=1, lr=1e-4, min_lr=5e-6, train_num_steps=1000, save_and_sample_every=1000,
                               log_freq=1, results_folder=str(res), resume_milestone=1, ema_update_after_step=1,
                               ema_update_every=1, warmup_iter=2),
               "data": dict(class_name="synthetic", batch_size=2)})
    dpm = build().to(dev).train()
    tr = Trainer(dpm, ImageStream(cfg.data, 2, (16, 16), dev, 0), cfg, dev, 0, 1)
    ck = torch.load(CKPT, map_location="cpu", weights_only=True)
    assert tr.step == 6 and tr.opt.step_count == int(g["opt.step0"]) == 6 and tr.ema_step == 6 and tr.ema_initted
    assert int(g["n_params"]) == len(tr.flat.params)
    close(tr.opt.m[:16], g["opt.exp_avg0"], rtol=0, atol=0)
    names = [n for n, p in dpm.named_parameters() if p.requires_grad]
    for n, p, o in list(zip(names, tr.flat.params, tr.flat.offsets))[::7]:
        assert torch.equal(p.detach().cpu(), ck["model"][n]), n
        assert torch.equal(tr.opt.ema[o:o + p.numel()].view(p.shape).cpu(), ck["ema"]["ema_model." + n]), n
        idx = names.index(n)
        assert torch.equal(tr.opt.v[o:o + p.numel()].view(p.shape).cpu(), ck["opt"]["state"][idx]["exp_avg_sq"]), n
    # the reference's 7TH STEP (train_uncond_dpm.py:264-310, ddm/ema.py:141-188) on the recorded draws: same image batch, the t and
    # the noise the reference drew (g13: s7.t, s7.noise), one optimiser step of the resumed trainer -> the reference's loss, its
    # pre-clip gradient norm, its learning rate, 16 parameters and their EMA shadows after opt.step() / ema.update()
    x0 = fill.hash_tensor((2, 3, 16, 16), "ck.x0", 1.0).to(dev)
    t7, n7 = torch.from_numpy(g["s7.t"]).to(dev), torch.from_numpy(g["s7.noise"]).to(dev)
    seen = {}
    orig_step = dpm.training_step

    def step_on_recorded_draws(batch):
        loss, log = orig_step(batch, t=t7, noise=n7)
        seen.update(loss=float(loss.detach()), simple=float(log["train/loss_simple"]))
        return loss, log

    dpm.training_step = step_on_recorded_draws
    tr.stream = iter(lambda: {"image": x0}, None)
    lr7 = tr.lr * tr._lr_ratio(tr.step)
    assert abs(lr7 - float(g["s7.lr"])) <= 1e-9 * float(g["s7.lr"]), (lr7, float(g["s7.lr"]))
    before = {n: dict(dpm.named_parameters())[n].detach().reshape(-1)[:16].cpu().clone() for n in g["s7.names"]}
    tr.train(max_steps=1)
    del dpm.training_step
    assert tr.step == 7 and tr.opt.step_count == 7 and tr.ema_step == int(g["s7.ema_step"]) == 7
    assert abs(seen["loss"] - float(g["s7.loss"])) <= 1e-4 * float(g["s7.loss"]), (seen["loss"], float(g["s7.loss"]))
    assert abs(seen["simple"] - float(g["s7.loss_simple"])) <= 1e-4 * float(g["s7.loss_simple"])
    assert abs(tr.opt.grad_norm() - float(g["s7.grad_norm"])) <= 1e-4 * float(g["s7.grad_norm"]), (tr.opt.grad_norm(), float(g["s7.grad_norm"]))
    params = dict(dpm.named_parameters())
    for i, n in enumerate(g["s7.names"]):
        n = str(n)
        o = tr.flat.offsets[names.index(n)]
        got_p = params[n].detach().reshape(-1)[:16].cpu()
        got_e = tr.opt.ema[o:o + 16].cpu()
        close(got_p, g["s7.param"][i], rtol=1e-5, atol=1e-7)
        close(got_e, g["s7.ema"][i], rtol=1e-5, atol=1e-7)
        # ... and the MOVE of the step itself (a resumed trainer that did nothing would pass the two lines above at lr 1e-4)
        want_d = torch.from_numpy(g["s7.param"][i]) - before[n]
        if float(want_d.abs().max()) > 0:
            assert float(((got_p - before[n]) - want_d).abs().max()) <= 2e-2 * float(want_d.abs().max()) + 2e-9, n
    tr.train(max_steps=1)                          # and it trains on from there
    assert tr.step == 8 and tr.opt.step_count == 8
