#!/usr/bin/env python3
"""Reference-FORMAT checkpoint fixture.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

Builds, with the reference's OWN classes (unet.uncond_unet_sd_2.EDMPrecond, ddm.ddm_const_2.DDPM, ddm.ema.EMA,
torch.optim.AdamW + LambdaLR exactly as /root/reference/train_uncond_dpm.py:169-189 constructs them), a tiny model, runs
six deterministic optimiser / EMA updates so that online != EMA and the optimiser has state, and writes the dict that
``Trainer.save`` (/root/reference/train_uncond_dpm.py:207-220) writes:

    {'step', 'model': model.state_dict(), 'opt': opt.state_dict(), 'lr_scheduler': ..., 'ema': ema.state_dict(), 'scaler': None}

to tests/golden/ref_checkpoint_model-1.pt (plain tensors / numbers only: loads with torch.load(weights_only=True)), plus
tests/golden/g13_checkpoint.npz = what the REFERENCE model produces from it: the 4-step deterministic sample from a fixed
x_T with the EMA weights (the path /root/reference/sample_uncond.py:131-147 takes) and with the online weights, and the
loss of one training step after resuming.  tests/test_hip_checkpoint.py feeds the same file to this build's loaders.
The fixture is data the reference wrote; no reference source is copied.
"""
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

from oracle import fill  # noqa: E402
import unet.uncond_unet_sd_2 as U  # noqa: E402  (the reference's)
import ddm.ddm_const_2 as D2  # noqa: E402
from ddm.ema import EMA  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
# 32 channels (a multiple of 32, the HIP path's channel granularity; below the 64 channels per head of the Dhariwal blocks, so the fixture has no attention layers (the file
# holds five copies of the weights -- model, EMA online + shadow, two AdamW moments -- and must stay small)
UNET_KW = dict(img_resolution=16, img_channels=3, model_type="DhariwalUNet", model_channels=32, channel_mult=[1],
               channel_mult_emb=2, num_blocks=1, attn_resolutions=[8], dropout=0.0, augment_dim=0)
MODEL_CFG = dict(eps=1e-3, sigma_max=1, sigma_min=0.001, weighting_loss=True, use_augment=False)


def build():
    torch.manual_seed(0)
    unet = U.EDMPrecond(**UNET_KW)
    dpm = D2.DDPM(model=unet, image_size=[16, 16], sampling_timesteps=4, loss_type="l2", start_dist="normal",
                  perceptual_weight=0.0, use_l1=False, cfg=dict(MODEL_CFG))
    sd = {k: (fill.fill_value("ck." + k, tuple(v.shape)) if v.is_floating_point() and v.dim() > 0 and "resample" not in k else v)
          for k, v in dpm.state_dict().items()}
    dpm.load_state_dict(sd)
    # run the reference p_losses un-modified with LPIPS == 0 (VGG16 weights are not fetchable; with perceptual_weight 0 the
    # reference crashes at ddm_const_2.py:251).  A plain attribute, not a sub-module: the state_dict is unchanged.
    dpm.perceptual_weight = 1.0
    dpm.perceptual_loss = lambda a, b: torch.zeros(a.shape[0], 1, 1, 1)
    return dpm


def main():
    dpm = build()
    lr, min_lr, steps = 1e-4, 5e-6, 1000

    def warm(it, warmup_iter=2):
        if it <= warmup_iter:
            return (it + 1) / warmup_iter
        return max((1 - (it - warmup_iter) / steps) ** 0.96, min_lr / lr)

    opt = torch.optim.AdamW(filter(lambda p: p.requires_grad, dpm.parameters()), lr=lr, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=warm)
    ema = EMA(dpm, ema_model=None, beta=0.9996, update_after_step=1, update_every=1)
    x0 = fill.hash_tensor((2, 3, 16, 16), "ck.x0", 1.0)
    for it in range(6):
        torch.manual_seed(100 + it)
        loss, _ = dpm.training_step({"image": x0})          # the reference draws t and the noise itself
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(dpm.parameters(), 1.0)
        opt.step(); sched.step(); ema.update()
    data = {"step": 6, "model": dpm.state_dict(), "opt": opt.state_dict(), "lr_scheduler": sched.state_dict(),
            "ema": ema.state_dict(), "scaler": None}
    path = os.path.join(OUT, "ref_checkpoint_model-1.pt")
    torch.save(data, path)
    back = torch.load(path, map_location="cpu", weights_only=True)         # the safe loader must accept it
    assert set(back) == set(data)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB; ema step {int(back['ema']['step'])}, initted {bool(back['ema']['initted'])}")

    # what the reference does with it (sample_uncond.py:131-147): strip 'ema_model.' and load into the model
    g = {}
    xT = fill.hash_tensor((2, 3, 16, 16), "ck.xT", 1.7, torch.float64)
    for tag, sd in (("ema", {k[10:]: v for k, v in back["ema"].items() if k.startswith("ema_model.")}),
                    ("online", back["model"])):
        m = build().eval()
        m.load_state_dict(sd)
        torch.manual_seed(7)
        orig = torch.randn
        torch.randn = lambda *a, **k: xT.clone().to(k.get("dtype", torch.float64))       # inject x_T into sample_fn_d
        try:
            with torch.no_grad():
                img = m.sample(batch_size=2)
        finally:
            torch.randn = orig
        g[tag + ".img"] = img.numpy()
        g[tag + ".w0"] = sd["model.model.enc.16x16_conv.weight"].reshape(-1)[:16].numpy()
    assert np.abs(g["ema.img"] - g["online.img"]).max() > 1e-6, "EMA and online weights must differ in the fixture"
    # the reference's 7TH STEP, continued from the live objects the checkpoint was written from (Trainer.train,
    # /root/reference/train_uncond_dpm.py:264-310: loss -> backward -> clip_grad_norm_(1.0) -> opt.step -> lr_scheduler.step ->
    # ema.update, /root/reference/ddm/ema.py:141-188).  The reference draws t and the noise itself (ddm_const_2.py:163-170,
    # 199-201); the same two draws are replayed here from the same seed and recorded, and the replay is checked against the
    # reference's own call.
    torch.manual_seed(106)
    t7 = torch.rand(2) * (1.0 - float(dpm.eps)) + float(dpm.eps)
    n7 = torch.randn_like(x0)
    torch.manual_seed(106)
    loss7, log7 = dpm.training_step({"image": x0})
    opt.zero_grad()
    loss7.backward()
    lr7 = float(opt.param_groups[0]["lr"])
    norm7 = float(torch.nn.utils.clip_grad_norm_(dpm.parameters(), 1.0))
    opt.step(); sched.step(); ema.update()
    names = [n for n, p in dpm.named_parameters() if p.requires_grad and p.numel() >= 16]
    pick = names[::max(1, len(names) // 16)][:16]
    sd7, ema7 = dpm.state_dict(), ema.state_dict()
    g["s7.t"], g["s7.noise"] = t7.numpy(), n7.numpy()
    g["s7.loss"] = np.array(float(loss7.detach()))
    g["s7.loss_simple"] = np.array(float(log7["train/loss_simple"]))
    g["s7.grad_norm"], g["s7.lr"] = np.array(norm7), np.array(lr7)
    g["s7.names"] = np.array(pick)
    g["s7.param"] = np.stack([sd7[n].reshape(-1)[:16].numpy() for n in pick])           # 16 parameters x their first 16 entries
    g["s7.ema"] = np.stack([ema7["ema_model." + n].reshape(-1)[:16].numpy() for n in pick])
    g["s7.ema_step"] = np.array(int(ema7["step"]))
    g["s7.ema_decay"] = np.array(float(ema.get_current_decay()))
    # the replayed draws are the reference's: p_losses on them reproduces the loss of the seeded call bit for bit
    m = build(); m.load_state_dict(back["model"])
    xn = m.q_sample(x_start=x0, noise=n7, t=t7, C=-x0)
    with torch.no_grad():
        cp, npred = m.model(xn, t7)
    w1, w2 = ((t7 - 1) / t7) ** 2 + 1, (t7 / (1 - t7 + m.eps)) ** 2 + 1
    chk = ((w1 * ((cp + x0) ** 2).sum([1, 2, 3]) + w2 * ((npred - n7) ** 2).sum([1, 2, 3])).sum() / 2)
    assert abs(float(chk) - float(loss7.detach())) <= 1e-5 * abs(float(loss7.detach())), (float(chk), float(loss7.detach()))
    print(f"7th step: loss {float(loss7.detach()):.6f}, grad norm {norm7:.6f}, lr {lr7:.4e}, ema step {int(ema7['step'])}, "
          f"decay {float(ema.get_current_decay()):.6f}")
    g["opt.exp_avg0"] = back["opt"]["state"][0]["exp_avg"].reshape(-1)[:16].numpy()
    g["opt.step0"] = np.array(float(back["opt"]["state"][0]["step"]))
    g["n_params"] = np.array(len(back["opt"]["state"]))
    np.savez_compressed(os.path.join(OUT, "g13_checkpoint.npz"), **g)
    print("wrote g13_checkpoint.npz", {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
