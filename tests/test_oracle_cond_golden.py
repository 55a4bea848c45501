"""CPU: the conditional-SR oracle (oracle/cond_unet_ref.py) reproduces the vectors the imported reference produced
(tools/make_golden_cond.py -> tests/golden/g14_cond_unet.npz, g15_cond_blocks.npz), and the sliding-window stitching
restatement has the properties of /root/reference/sample_cond_ldm.py:281-330."""
import os

import numpy as np
import torch

from oracle import cond_unet_ref as R
from oracle import fill

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def close(a, b, rtol=1e-3, atol=1e-4):
    a = torch.as_tensor(np.asarray(a.detach() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    s = max(float(b.abs().max()), 1e-12)
    torch.testing.assert_close(a / s, b / s, rtol=rtol, atol=atol)


def test_cond_unet_reduced_width_eval_and_train():
    g = np.load(os.path.join(G, "g14_cond_unet.npz"))
    cfg = R.default_cfg(dim=32)
    sd = R.filled_state_dict(cfg)
    x = fill.hash_tensor((2, 3, 32, 32), "cond.x", 1.0)
    tt = torch.tensor([0.3, 0.85])
    hm = R.cond_features(2, 32, 32)
    gx, gy = fill.hash_tensor((2, 3, 32, 32), "cond.gx", 1.0), fill.hash_tensor((2, 3, 32, 32), "cond.gy", 1.0)
    for mode in ("eval", "train"):
        sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "time_mlp.0.W" else v.clone())
               for k, v in sd.items()}
        upd = {}
        o1, o2 = R.unet_forward(sdo, cfg, x, tt, hm, training=(mode == "train"), bn_update=upd)
        close(o1, g[f"{mode}.x1"]); close(o2, g[f"{mode}.x2"])
        ((o1 * gx).sum() + (o2 * gy).sum()).backward()
        for key in g.files:
            if key.startswith(f"{mode}.grad."):
                close(sdo[key[len(mode) + 6:]].grad.reshape(-1)[:4096], g[key])
        if mode == "train":
            close(upd["relation_layers_down.0.input_conv2.1.running_var"], g["train.bn.relation_layers_down.0.input_conv2.1.running_var"])
            close(upd["relation_layers_up.1.input_conv1.1.running_mean"], g["train.bn.relation_layers_up.1.input_conv1.1.running_mean"])


def test_two_decoder_variant_vs_reference_golden():
    """unet.cond_unet.Unet (two decoders, the class the DIV2K YAML names): tools/make_golden_cond2.py imported the reference
    module (pytorch_lightning.LightningModule -> nn.Module at import time) and recorded its outputs and gradients (g16)."""
    g = np.load(os.path.join(G, "g16_cond_unet_two_decoders.npz"))
    cfg = R.default_cfg(dim=32, two_decoders=True)
    sd = R.filled_state_dict(cfg)
    names = list(sd)
    assert names.index("ups.0.0.mlp.1.weight") < names.index("relation_layers_up.0.input_conv1.0.weight") < names.index("ups2.0.0.mlp.1.weight") \
        < names.index("relation_layers_up2.0.input_conv1.0.weight")            # registration order of cond_unet.py:710-714
    x = fill.hash_tensor((2, 3, 32, 32), "cond.x", 1.0)
    tt = torch.tensor([0.3, 0.85])
    hm = R.cond_features(2, 32, 32)
    gx, gy = fill.hash_tensor((2, 3, 32, 32), "cond.gx", 1.0), fill.hash_tensor((2, 3, 32, 32), "cond.gy", 1.0)
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "time_mlp.0.W" else v.clone())
           for k, v in sd.items()}
    o1, o2 = R.unet_forward(sdo, cfg, x, tt, hm, training=False)
    close(o1, g["eval.x1"]); close(o2, g["eval.x2"])
    ((o1 * gx).sum() + (o2 * gy).sum()).backward()
    for key in g.files:
        if key.startswith("eval.grad."):
            close(sdo[key[10:]].grad.reshape(-1)[:4096], g[key])


def test_param_shapes_cover_both_variants():
    one = R.param_shapes(R.default_cfg())
    two = R.param_shapes(R.default_cfg(two_decoders=True))
    assert set(one) < set(two)
    extra = {k.split(".")[0] for k in set(two) - set(one)}
    assert extra == {"ups2", "relation_layers_up2", "decouple2", "final_res_block2", "final_conv2"}      # cond_unet.py:713-785
    n = sum(int(np.prod(s)) for k, s in one.items() if "running" not in k and "num_batches" not in k)
    assert n == 87418440          # the DIV2K recipe's denoiser without its Swin-B encoder


def test_sliding_window_stitching_properties():
    """Overlap-averaging of per-window outputs: a constant field stays constant, window origins follow the reference's
    clamp-to-the-border rule, every output pixel is covered, and `ori_size` crops."""
    wins = R.slide_windows(200, 136, (128, 128), (64, 64))
    assert wins == [(0, 128, 0, 128), (0, 128, 8, 136), (64, 192, 0, 128), (64, 192, 8, 136), (72, 200, 0, 128), (72, 200, 8, 136)]
    assert R.slide_windows(100, 100, (128, 128), (64, 64)) == [(0, 100, 0, 100)]      # smaller than the crop: one window
    cond = fill.hash_tensor((1, 3, 40, 24), "sw.c", 1.0)
    calls = []

    def fn(c):
        calls.append(tuple(c.shape))
        return torch.nn.functional.interpolate(c, scale_factor=4, mode="nearest") * 2.0 + 1.0

    out = R.slide_sample_sr(fn, cond, (160, 96), (16, 16), (8, 8), ori_size=(150, 90))
    assert out.shape == (1, 3, 150, 90) and len(calls) == 4 * 2 and all(s == (1, 3, 16, 16) for s in calls)
    want = (torch.nn.functional.interpolate(cond, scale_factor=4, mode="nearest") * 2.0 + 1.0)[:, :, :150, :90]
    torch.testing.assert_close(out, want, rtol=1e-6, atol=1e-6)      # windows agree where they overlap -> the mean is the field
