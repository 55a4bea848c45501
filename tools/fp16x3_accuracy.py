#!/usr/bin/env python3
"""CPU study (numpy): an f32 product from THREE fp16 products (two-term round-to-nearest split, per-tensor power-of-two scale)
against the shipped six-bf16-product scheme and plain f32, error vs fp64 of dot products of length K relative to sum |a b|.
Not part of the product: evidence for DESIGN.md section 6 (what the next version of the split kernels could run on)."""
import numpy as np

rng = np.random.default_rng(0)


def split_bf16x3(a):
    a = a.astype(np.float32)
    def top(x):
        return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    a0 = top(a); r = a - a0; a1 = top(r); a2 = r - a1
    return a0, a1, top(a2)


def dot_bf16x6(a, b):
    a0, a1, a2 = split_bf16x3(a); b0, b1, b2 = split_bf16x3(b)
    f = lambda x, y: (x.astype(np.float64) * y.astype(np.float64)).sum(-1)     # products exact, accumulation idealised
    return f(a0, b0) + f(a0, b1) + f(a1, b0) + f(a1, b1) + f(a0, b2) + f(a2, b0)


def split_fp16x2(a, scale):
    s = (a.astype(np.float32) * np.float32(scale))
    a0 = s.astype(np.float16)
    a1 = (s - a0.astype(np.float32)).astype(np.float16)
    return a0, a1


def pow2_scale(a):
    m = float(np.abs(a).max())
    return 2.0 ** np.floor(np.log2(32768.0 / m))       # max lands in [16384, 32768): below fp16's 65504 with a x4 Winograd margin


def dot_fp16x3(a, b):
    sa, sb = pow2_scale(a), pow2_scale(b)
    a0, a1 = split_fp16x2(a, sa); b0, b1 = split_fp16x2(b, sb)
    f = lambda x, y: (x.astype(np.float64) * y.astype(np.float64)).sum(-1)
    return (f(a0, b0) + f(a0, b1) + f(a1, b0)) / (sa * sb)


def dot_f32(a, b):
    return np.cumsum((a.astype(np.float32) * b.astype(np.float32)), axis=-1, dtype=np.float32)[..., -1].astype(np.float64)


def report(name, a, b):
    ref = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
    den = (np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64))).sum(-1)
    out = []
    for tag, fn in (("f32 fma chain", dot_f32), ("6 x bf16", dot_bf16x6), ("3 x fp16", dot_fp16x3)):
        e = np.abs(fn(a, b) - ref) / den
        out.append(f"{tag}: max {e.max():.2e} mean {e.mean():.2e}")
    print(f"{name:46s} " + " | ".join(out))


K, R = 3456, 2048
a = rng.standard_normal((R, K)).astype(np.float32); b = (rng.standard_normal((R, K)) * 0.02).astype(np.float32)
report("normal x normal", a, b)
report("all-positive", np.abs(a), np.abs(b))
g = (rng.standard_normal((R, K)) * np.exp(rng.standard_normal((R, K)) * 3.0) * 1e-6).astype(np.float32)      # heavy-tailed gradients
report("heavy-tailed (log-normal sigma 3) x normal", g, a)
o = a.copy(); o[:, 0] = 3e4                                                                                      # one huge outlier per tensor
report("one outlier 3e4 x normal", o, b)
t = (a * 1e-30).astype(np.float32)
report("tiny magnitudes (1e-30) x normal", t, b)
