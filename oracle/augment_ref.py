"""Oracle (CPU, plain PyTorch) restatement of the augmentation pipe the DDM wrappers apply to x_start when
``use_augment: True`` (/root/reference/ddm/augment.py:119-327, instantiated at ddm_const.py:179-180 with p=0.15 and
ddm_const_2.py:112-113 with p=0.12, both with xflip=1e8, yflip=1, scale=1, rotate_frac=1, aniso=1, translate_frac=1).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity PINNED by tools/make_golden_augment.py against the imported
reference class with its RNG calls replaced by a recorded stream -> tests/golden/g12_augment.npz.

Only the six transforms DDM enables are restated (pixel blits: x/y flip; geometric: isotropic scale, rotation,
anisotropic scale, translation; 9 label columns).  Every random draw is an explicit input (``draws``), in the order
the reference consumes them:
  xflip_bit [N] in {0,1}, xflip_u [N];  yflip_bit, yflip_u;  scale_n [N] ~N(0,1), scale_u;  rot_v [N] ~U(0,1), rot_u;
  aniso_n ~N, aniso_r ~U, aniso_u, aniso_ru;  trans_n [2,N] ~N, trans_u [N]
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
        0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
        0.0017677118642428036, -0.007800708325034148]        # sym6 low-pass decomposition filter (augment.py:37)

DRAW_KEYS = ("xflip_bit", "xflip_u", "yflip_bit", "yflip_u", "scale_n", "scale_u", "rot_v", "rot_u", "aniso_n", "aniso_r",
             "aniso_u", "aniso_ru", "trans_n", "trans_u")


def _mat(rows, like):
    """[N,3,3] from a 3x3 nest of python numbers / [N] tensors."""
    cols = [[(e if isinstance(e, torch.Tensor) else torch.full_like(like, float(e))) for e in r] for r in rows]
    return torch.stack([torch.stack(r, dim=-1) for r in cols], dim=-2)


def _T(tx, ty, like):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], like)


def _S(sx, sy, like):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], like)


def _R(th, like):
    return _mat([[torch.cos(th), torch.sin(-th), 0], [torch.sin(th), torch.cos(th), 0], [0, 0, 1]], like)


def gates_and_labels(draws: Dict[str, torch.Tensor], p: float, xflip=1e8, yflip=1.0, scale=1.0, rotate_frac=1.0, aniso=1.0,
                     translate_frac=1.0, scale_std=0.2, rotate_frac_max=1.0, aniso_std=0.2, aniso_rotate_prob=0.5):
    """Per-image parameters after gating, and the [N, 9] label matrix (augment.py:161-172, 202-232, 326)."""
    d = draws
    z = lambda t: torch.zeros_like(t)
    fx = torch.where(d["xflip_u"] < xflip * p, d["xflip_bit"], z(d["xflip_bit"]))
    fy = torch.where(d["yflip_u"] < yflip * p, d["yflip_bit"], z(d["yflip_bit"]))
    ws = torch.where(d["scale_u"] < scale * p, d["scale_n"], z(d["scale_n"]))
    wr = (d["rot_v"] * 2 - 1) * (math.pi * rotate_frac_max)
    wr = torch.where(d["rot_u"] < rotate_frac * p, wr, z(wr))
    wa = torch.where(d["aniso_u"] < aniso * p, d["aniso_n"], z(d["aniso_n"]))
    ra = (d["aniso_r"] * 2 - 1) * math.pi
    ra = torch.where(d["aniso_ru"] < aniso_rotate_prob, ra, z(ra))
    wt = torch.where(d["trans_u"][None] < translate_frac * p, d["trans_n"], z(d["trans_n"]))
    labels = torch.stack([fx.float(), fy.float(), ws, wr.cos() - 1, wr.sin(), wa * ra.cos(), wa * ra.sin(), wt[0], wt[1]], dim=1)
    return dict(fx=fx, fy=fy, ws=ws, wr=wr, wa=wa, ra=ra, wt=wt), labels


def inverse_transform(g, H, W, scale_std=0.2, aniso_std=0.2, translate_frac_std=0.125):
    """G_inv [N,3,3] in pixel units about the image centre (augment.py:199-232)."""
    like = g["ws"]
    s = (g["ws"] * scale_std).exp2()
    G = _S(1 / s, 1 / s, like)                                    # eye @ scale2d_inv(s, s)
    G = G @ _R(g["wr"], like)                                     # rotate2d_inv(-w) = rotate2d(w)
    sa = (g["wa"] * aniso_std).exp2()
    G = G @ _R(-g["ra"], like) @ _S(1 / sa, 1 / (1 / sa), like) @ _R(g["ra"], like)     # scale2d_inv(s, 1/s)
    G = G @ _T(-g["wt"][0] * (W * translate_frac_std), -g["wt"][1] * (H * translate_frac_std), like)
    return G


def margins(G, H, W, hz_pad=3):
    """Reflect-padding amounts [mx0, my0, mx1, my1] (ints, batch-wide) so that the warped image corners stay inside
    the padded image plus the filter footprint (augment.py:239-248)."""
    cx, cy = (W - 1) / 2, (H - 1) / 2
    cp = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=G.dtype)
    q = G @ cp.t()                                                # [N, 3, 4]
    m = q[:, :2, :].permute(1, 0, 2).flatten(1)                   # [2, N*4]
    m = torch.cat([-m, m]).max(dim=1).values                      # x0, y0, x1, y1
    m = m + torch.tensor([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, dtype=G.dtype)
    m = m.max(torch.zeros(4, dtype=G.dtype)).min(torch.tensor([W - 1, H - 1] * 2, dtype=G.dtype))
    return m.ceil().to(torch.int32)


def _conv1d(x, w, dim, stride=1, pad=0):
    """depth-wise correlation along one spatial dim of NCHW x with the 1-D kernel w."""
    C = x.shape[1]
    k = torch.as_tensor(w, dtype=x.dtype)
    if dim == 3:
        return F.conv2d(x, k.reshape(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C, stride=(1, stride), padding=(0, pad))
    return F.conv2d(x, k.reshape(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C, stride=(stride, 1), padding=(pad, 0))


def _zero_stuff(x, dim):
    """interleave zeros after every sample along dim and drop the trailing one: length L -> 2L - 1."""
    y = torch.stack([x, torch.zeros_like(x)], dim=dim + 1)
    shape = list(x.shape)
    shape[dim] *= 2
    y = y.reshape(shape)
    return y.narrow(dim, 0, shape[dim] - 1)


def warp(images, G, m):
    """Reflect-pad, x2 up-sample (sym6), bilinear affine resample, x2 down-sample (sym6), crop (augment.py:250-276)."""
    N, C, H, W = images.shape
    like = G[:, 0, 0]
    hz = torch.tensor(SYM6, dtype=torch.float32)
    hp = len(SYM6) // 4
    mx0, my0, mx1, my1 = (int(v) for v in m)
    x = F.pad(images, [mx0, mx1, my0, my1], mode="reflect")
    G = _T((mx0 - mx1) / 2, (my0 - my1) / 2, like) @ G
    up_pad = (len(SYM6) + 1) // 2
    x = _conv1d(_zero_stuff(x, 3), hz.flip(0), 3, pad=up_pad)
    x = _conv1d(_zero_stuff(x, 2), hz.flip(0), 2, pad=up_pad)
    G = _S(2, 2, like) @ G @ _S(0.5, 0.5, like)
    G = _T(-0.5, -0.5, like) @ G @ _T(0.5, 0.5, like)
    shape = [N, C, (H + hp * 2) * 2, (W + hp * 2) * 2]
    G = _S(2 / x.shape[3], 2 / x.shape[2], like) @ G @ _S(1 / (2 / shape[3]), 1 / (2 / shape[2]), like)
    grid = F.affine_grid(theta=G[:, :2, :], size=shape, align_corners=False)
    x = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    dn_pad = (len(SYM6) - 1) // 2
    x = _conv1d(x, hz, 3, stride=2, pad=dn_pad)[:, :, :, hp:-hp]
    x = _conv1d(x, hz, 2, stride=2, pad=dn_pad)[:, :, hp:-hp, :]
    return x


def augment(images, draws: Dict[str, torch.Tensor], p: float):
    """images [N,C,H,W] -> (augmented images, labels [N,9])."""
    N, C, H, W = images.shape
    g, labels = gates_and_labels(draws, p)
    x = torch.where(g["fx"].reshape(N, 1, 1, 1) == 1, images.flip(3), images)
    x = torch.where(g["fy"].reshape(N, 1, 1, 1) == 1, x.flip(2), x)
    G = inverse_transform(g, H, W)
    return warp(x, G, margins(G, H, W)), labels


def make_draws(N: int, seed: int, force: float = 0.0) -> Dict[str, torch.Tensor]:
    """Deterministic draw set; ``force`` shifts the gate uniforms down so more transforms fire in small tests."""
    gen = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=gen)
    n = lambda *s: torch.randn(*s, generator=gen)
    b = lambda: torch.randint(2, (N,), generator=gen)
    gate = lambda: (u(N) - force).clamp(min=0.0)
    return {"xflip_bit": b(), "xflip_u": u(N), "yflip_bit": b(), "yflip_u": gate(), "scale_n": n(N), "scale_u": gate(),
            "rot_v": u(N), "rot_u": gate(), "aniso_n": n(N), "aniso_r": u(N), "aniso_u": gate(), "aniso_ru": u(N),
            "trans_n": n(2, N), "trans_u": gate()}
