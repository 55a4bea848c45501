#!/usr/bin/env python3
"""Pin for the TWO-DECODER conditional denoiser `unet.cond_unet.Unet` (the class the DIV2K YAML names;
/root/reference/unet/cond_unet.py:592-918).  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

Round 2 restated its extra wiring (second decoder, `relation_layers_up2`, `decouple2`, `final_res_block2` / `final_conv2`, the
second output's preconditioning, cond_unet.py:885-917) from the text because the module subclasses
`pytorch_lightning.LightningModule` and pytorch_lightning is not installed.  Nothing of Lightning is used by `Unet.__init__` /
`Unet.forward` (it is a base class only), so the module imports with the same IMPORT-TIME-ONLY placeholders
tools/make_golden_cond.py uses for torchvision / fvcore plus a `pytorch_lightning` placeholder whose `LightningModule` is
`torch.nn.Module`.  The condition encoder (Swin-B: torchvision ops + unfetchable weights) is never built; its four feature maps
are injected, as in g14.

Checks oracle.cond_unet_ref (two_decoders=True) against the imported reference on identical inputs -- both outputs and every
parameter gradient -- and writes tests/golden/g16_cond_unet_two_decoders.npz + oracle_vs_reference_report_cond2.json.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


class _Meta(type):
    def __getattr__(cls, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return cls()


class _Stub(metaclass=_Meta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return a[0] if (len(a) == 1 and callable(a[0]) and not k) else self

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Stub()


class _StubModule(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Stub


for name in ["torchvision", "torchvision.ops", "torchvision.ops.misc", "torchvision.ops.stochastic_depth", "torchvision.transforms",
             "torchvision.transforms._presets", "torchvision.utils", "torchvision.models", "torchvision.models._api",
             "torchvision.models._meta", "torchvision.models._utils", "fvcore", "fvcore.common", "fvcore.common.config"]:
    m = _StubModule(name); m.__path__ = []; sys.modules[name] = m


class CfgNode(dict):
    __getattr__ = dict.get


sys.modules["fvcore.common.config"].CfgNode = CfgNode
pl = types.ModuleType("pytorch_lightning")
pl.LightningModule = nn.Module                      # base class only: Unet uses none of Lightning's methods
sys.modules["pytorch_lightning"] = pl

import unet.cond_unet as C2  # noqa: E402  (the reference's two-decoder module)
import unet.swin_transformer as S  # noqa: E402

S.swin_b = lambda weights=None: None

from oracle import cond_unet_ref as R  # noqa: E402
from oracle import fill  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)
report = {"torch": torch.__version__, "module": "unet.cond_unet", "cases": []}


def check(name, got, want, tol=2e-5):
    a, b = got.double(), want.double()
    e = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    report["cases"].append(dict(case=name, max_rel_err=e, tol=tol, ok=bool(e <= tol)))
    print(f"{'OK ' if e <= tol else 'BAD'} {name}: rel_err={e:.3e}")
    assert e <= tol, name


cfg = R.default_cfg(dim=32, two_decoders=True)
m = C2.Unet(dim=cfg["dim"], dim_mults=tuple(cfg["dim_mults"]), cond_dim=cfg["dim"], cond_dim_mults=(), channels=cfg["channels"],
            cond_in_dim=3, window_sizes1=cfg["window_sizes1"], window_sizes2=cfg["window_sizes2"], fourier_scale=cfg["fourier_scale"],
            cfg=CfgNode({"cond_pe": False, "cond_net": "swin"}), cond_net="swin", cond_pe=False)
shapes = R.param_shapes(cfg)
ref = {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("init_conv_mask")}
assert ref == {k: tuple(s) for k, s in shapes.items()}, (sorted(set(ref) ^ set(shapes))[:10],
                                                        [k for k in ref if k in shapes and ref[k] != tuple(shapes[k])][:5])
assert list(ref) == list(shapes), "registration order differs"
report["cases"].append(dict(case="G16/state_dict names, shapes and order", max_rel_err=0.0, tol=0.0, ok=True))
sd = R.filled_state_dict(cfg)
missing, unexpected = m.load_state_dict(sd, strict=False)
assert not unexpected and all(k.startswith("init_conv_mask") for k in missing), (missing[:5], unexpected[:5])
for mod in m.modules():
    if isinstance(mod, nn.Dropout):
        mod.p = 0.0

g16 = {}
B, H = 2, 32
x = fill.hash_tensor((B, 3, H, H), "cond.x", 1.0)
tt = torch.tensor([0.3, 0.85])
hm = R.cond_features(B, H, H)
m.init_conv_mask = lambda mask: [h.clone() for h in hm]
gx, gy = fill.hash_tensor((B, 3, H, H), "cond.gx", 1.0), fill.hash_tensor((B, 3, H, H), "cond.gy", 1.0)
for mode in ("eval", "train"):
    m.train(mode == "train")
    m.load_state_dict(sd, strict=False)
    m.zero_grad()
    y1, y2 = m(x, tt, None)
    ((y1 * gx).sum() + (y2 * gy).sum()).backward()
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "time_mlp.0.W" else v.clone())
           for k, v in sd.items()}
    o1, o2 = R.unet_forward(sdo, cfg, x, tt, hm, training=(mode == "train"), bn_update={})
    ((o1 * gx).sum() + (o2 * gy).sum()).backward()
    check(f"G16/{mode}/x1", o1, y1); check(f"G16/{mode}/x2", o2, y2)
    g16[f"{mode}.x1"], g16[f"{mode}.x2"] = y1.detach().numpy(), y2.detach().numpy()
    named = dict(m.named_parameters())
    gmax = max(float(p.grad.double().norm()) for k, p in named.items() if p.grad is not None)
    worst, n = 0.0, 0
    for k, p in named.items():
        if k.startswith("init_conv_mask") or p.grad is None:
            continue
        gn_ref = float(p.grad.double().norm())
        e = float((sdo[k].grad.double() - p.grad.double()).norm()) / (gn_ref + 1e-4 * gmax)
        worst, n = max(worst, e), n + 1
        g16[f"{mode}.gradnorm.{k}"] = np.array(gn_ref)
    g16[f"{mode}.gradnorm_max"] = np.array(gmax)
    report["cases"].append(dict(case=f"G16/{mode}/all-grad-norms ({n} parameters)", max_rel_err=worst, tol=1e-4, ok=bool(worst < 1e-4)))
    print(f"G16/{mode}: worst gradient rel err over {n} parameters {worst:.2e}")
    assert worst < 1e-4
    for k in ("ups2.0.0.block1.proj.weight", "ups2.3.3.weight", "relation_layers_up2.1.attentions.0.q_lin.weight", "decouple2.1.weight",
              "final_res_block2.block2.proj.weight", "final_conv2.weight", "ups.1.3.1.weight", "downs.0.0.block1.proj.weight"):
        check(f"G16/{mode}/grad/{k}", sdo[k].grad, named[k].grad, 1e-4)
        g16[f"{mode}.grad.{k}"] = named[k].grad.reshape(-1)[:4096].numpy().copy()
np.savez_compressed(os.path.join(OUT, "g16_cond_unet_two_decoders.npz"), **g16)
report["max_rel_err"] = max(c["max_rel_err"] for c in report["cases"])
report["n_cases"] = len(report["cases"])
json.dump(report, open(os.path.join(OUT, "oracle_vs_reference_report_cond2.json"), "w"), indent=1)
print(f"ALL OK: {report['n_cases']} cases, max rel err {report['max_rel_err']:.3e}")
