#!/usr/bin/env python3
"""Golden vectors for the augmentation pipe (use_augment: True).  RUNS ONLY IN THE BUILD CONTAINER (needs
/root/reference); same import shims as tools/make_golden.py.  The reference AugmentPipe (ddm/augment.py) draws from
torch.randint / rand / randn; those three are replaced, for the duration of each call, by functions that hand out a
recorded stream (oracle.augment_ref.make_draws) in the order the reference consumes it, so the oracle restatement and
the HIP implementation can be fed the very same draws.  Writes tests/golden/g12_augment.npz + a report."""
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

from oracle import augment_ref as A, fill  # noqa: E402

from ddm.augment import AugmentPipe  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
report = {"torch": torch.__version__, "cases": []}


def check(name, got, want, tol=2e-5):
    e = float((got.double() - want.double()).abs().max() / (want.double().abs().max() + 1e-30))
    report["cases"].append(dict(case=name, max_rel_err=e, tol=tol, ok=bool(e <= tol)))
    print(f"{'OK ' if e <= tol else 'BAD'} {name}: rel_err={e:.3e}")
    assert e <= tol, name


def run_reference(pipe, images, d):
    ints = iter([d["xflip_bit"], d["yflip_bit"]])
    unis = iter([d["xflip_u"], d["yflip_u"], d["scale_u"], d["rot_v"], d["rot_u"], d["aniso_r"], d["aniso_u"], d["aniso_ru"],
                 d["trans_u"]])
    nrms = iter([d["scale_n"], d["aniso_n"], d["trans_n"]])
    o = (torch.randint, torch.rand, torch.randn)
    torch.randint = lambda hi, shape, **k: next(ints).reshape(shape).clone()
    torch.rand = lambda shape, **k: next(unis).reshape(shape).clone()
    torch.randn = lambda shape, **k: next(nrms).reshape(shape).clone()
    try:
        out = pipe(images)
    finally:
        torch.randint, torch.rand, torch.randn = o
    for it in (ints, unis, nrms):
        assert next(it, None) is None, "the reference consumed fewer draws than recorded"
    return out


g = {}
kw = dict(xflip=1e8, yflip=1, scale=1, rotate_frac=1, aniso=1, translate_frac=1)
for tag, p, N, H, W, seed, force in (("p012", 0.12, 16, 32, 32, 11, 0.0), ("p015", 0.15, 16, 32, 32, 12, 0.0),
                                      ("forced", 0.12, 8, 32, 32, 13, 0.9), ("forced64", 0.15, 4, 64, 64, 14, 0.9),
                                      ("identity", 0.12, 4, 32, 32, 15, -2.0)):
    pipe = AugmentPipe(p=p, **kw)
    x = fill.hash_tensor((N, 3, H, W), f"aug.{tag}.x", 1.0)
    d = A.make_draws(N, seed, force)
    y_ref, lab_ref = run_reference(pipe, x, d)
    y_o, lab_o = A.augment(x, d, p)
    assert lab_ref.shape == (N, 9) and y_ref.shape == x.shape
    check(f"G12/{tag}/labels", lab_o, lab_ref, 1e-6)
    check(f"G12/{tag}/images", y_o, y_ref)
    fired = int((lab_ref[:, 1:] != 0).any(dim=1).sum())
    report["cases"][-1]["images_with_a_geometric_or_yflip_transform"] = fired
    g[f"{tag}.images"] = y_ref.numpy(); g[f"{tag}.labels"] = lab_ref.numpy()
np.savez_compressed(os.path.join(OUT, "g12_augment.npz"), **g)
report["all_ok"] = all(c["ok"] for c in report["cases"])
with open(os.path.join(OUT, "oracle_vs_reference_report_augment.json"), "w") as f:
    json.dump(report, f, indent=1)
print("ALL OK", len(report["cases"]), "cases")
