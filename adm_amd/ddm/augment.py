"""AugmentPipe of the DDM wrappers (``use_augment: True``) on the HIP hot path.

Mirror of /root/reference/ddm/augment.py:119-327 for the transforms DDM switches on (ddm_const.py:179-180 with p = 0.15,
ddm_const_2.py:112-113 with p = 0.12: xflip=1e8, yflip=1, scale=1, rotate_frac=1, aniso=1, translate_frac=1):
``AugmentPipe(p=..., xflip=..., ...)(images) -> (images, labels [N, 9])``.

Split of work: the per-image parameters -- gates, the 3x3 inverse transform, the batch-wide reflect-padding margin, the
normalised affine ``theta`` and the nine label columns -- are O(N) scalars computed with a handful of tiny device-side
torch ops; the image work (flips, reflect pad, sym6 x2 up-sample, bilinear affine resample, sym6 x2 down-sample, crop)
is ``adm_augment_geometric`` (adm_amd/csrc/augment.hip).  Unlike the reference, which reads the padding margin back to
the host every call (``.ceil().to(int32)`` feeding ``F.pad``), nothing here synchronises with the device.

Pixel-blitting ``rotate_int`` / ``translate_int`` and the colour transforms exist in the reference class but are not
enabled by any DDM wrapper or config: asking for them raises.  ``draws=`` injects the random numbers (parity tests).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .. import hip
from ..hip import call, ptr


def _mat(rows, like):
    cols = [[(e if isinstance(e, torch.Tensor) else torch.full_like(like, float(e))) for e in r] for r in rows]
    return torch.stack([torch.stack(r, dim=-1) for r in cols], dim=-2)


def _T(tx, ty, like):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], like)


def _S(sx, sy, like):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], like)


def _R(th, like):
    return _mat([[torch.cos(th), torch.sin(-th), 0], [torch.sin(th), torch.cos(th), 0], [0, 0, 1]], like)


class AugmentPipe:
    def __init__(self, p=1, xflip=0, yflip=0, rotate_int=0, translate_int=0, translate_int_max=0.125, scale=0, rotate_frac=0,
                 aniso=0, translate_frac=0, scale_std=0.2, rotate_frac_max=1, aniso_std=0.2, aniso_rotate_prob=0.5,
                 translate_frac_std=0.125, brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, **unused_color_kw):
        if rotate_int or translate_int or brightness or contrast or lumaflip or hue or saturation:
            raise NotImplementedError("rotate_int / translate_int / colour transforms are not enabled by any DDM wrapper")
        self.p = float(p)
        self.xflip, self.yflip = float(xflip), float(yflip)
        self.scale, self.rotate_frac, self.aniso, self.translate_frac = float(scale), float(rotate_frac), float(aniso), float(translate_frac)
        self.scale_std, self.rotate_frac_max = float(scale_std), float(rotate_frac_max)
        self.aniso_std, self.aniso_rotate_prob = float(aniso_std), float(aniso_rotate_prob)
        self.translate_frac_std = float(translate_frac_std)
        self.label_dim = (self.xflip > 0) + (self.yflip > 0) + (self.scale > 0) + 2 * (self.rotate_frac > 0) + \
            2 * (self.aniso > 0) + 2 * (self.translate_frac > 0)

    def _draws(self, N, dev) -> Dict[str, torch.Tensor]:
        u = lambda *s: torch.rand(*s, device=dev)
        n = lambda *s: torch.randn(*s, device=dev)
        b = lambda: torch.randint(2, (N,), device=dev)
        return {"xflip_bit": b(), "xflip_u": u(N), "yflip_bit": b(), "yflip_u": u(N), "scale_n": n(N), "scale_u": u(N),
                "rot_v": u(N), "rot_u": u(N), "aniso_n": n(N), "aniso_r": u(N), "aniso_u": u(N), "aniso_ru": u(N),
                "trans_n": n(2, N), "trans_u": u(N)}

    @torch.no_grad()
    def __call__(self, images, draws: Optional[Dict[str, torch.Tensor]] = None):
        hip.require_cuda(images, "images")
        images = images.to(torch.float32).contiguous()
        N, C, H, W = images.shape
        dev = images.device
        d = self._draws(N, dev) if draws is None else {k: v.to(dev) for k, v in draws.items()}
        zero = torch.zeros(N, device=dev)
        zi = torch.zeros(N, device=dev, dtype=d["xflip_bit"].dtype)
        labels = []
        # ---- gates + labels (augment.py:161-172, 202-232) ----
        fx = torch.where(d["xflip_u"] < self.xflip * self.p, d["xflip_bit"], zi) if self.xflip > 0 else zi
        fy = torch.where(d["yflip_u"] < self.yflip * self.p, d["yflip_bit"], zi) if self.yflip > 0 else zi
        if self.xflip > 0:
            labels.append(fx.float())
        if self.yflip > 0:
            labels.append(fy.float())
        like = zero
        G = _S(1, 1, like)
        if self.scale > 0:
            w = torch.where(d["scale_u"] < self.scale * self.p, d["scale_n"], zero)
            s = (w * self.scale_std).exp2()
            G = G @ _S(1 / s, 1 / s, like)
            labels.append(w)
        if self.rotate_frac > 0:
            w = (d["rot_v"] * 2 - 1) * (math.pi * self.rotate_frac_max)
            w = torch.where(d["rot_u"] < self.rotate_frac * self.p, w, zero)
            G = G @ _R(w, like)
            labels += [w.cos() - 1, w.sin()]
        if self.aniso > 0:
            w = torch.where(d["aniso_u"] < self.aniso * self.p, d["aniso_n"], zero)
            r = (d["aniso_r"] * 2 - 1) * math.pi
            r = torch.where(d["aniso_ru"] < self.aniso_rotate_prob, r, zero)
            s = (w * self.aniso_std).exp2()
            G = G @ _R(-r, like) @ _S(1 / s, 1 / (1 / s), like) @ _R(r, like)
            labels += [w * r.cos(), w * r.sin()]
        if self.translate_frac > 0:
            w = torch.where(d["trans_u"][None] < self.translate_frac * self.p, d["trans_n"], torch.zeros_like(d["trans_n"]))
            G = G @ _T(-w[0] * (W * self.translate_frac_std), -w[1] * (H * self.translate_frac_std), like)
            labels += [w[0], w[1]]
        labels = torch.stack(labels, dim=1) if labels else torch.zeros(N, 0, device=dev)
        # ---- batch-wide reflect-padding margin, kept on the device (augment.py:239-248) ----
        cx, cy = (W - 1) / 2, (H - 1) / 2
        cp = torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], device=dev)
        q = G @ cp.t()
        m = q[:, :2, :].permute(1, 0, 2).flatten(1)
        m = torch.cat([-m, m]).max(dim=1).values
        m = m + torch.tensor([6 - cx, 6 - cy] * 2, device=dev)
        m = m.max(torch.zeros(4, device=dev)).min(torch.tensor([W - 1.0, H - 1.0] * 2, device=dev)).ceil()
        margin = m.to(torch.int32).contiguous()
        mx0, my0, mx1, my1 = m[0], m[1], m[2], m[3]               # 0-dim device tensors
        # ---- theta of affine_grid (augment.py:251-266), with the margin-dependent factors as device scalars ----
        e = lambda v: v.expand(N)
        G = _T(e((mx0 - mx1) / 2), e((my0 - my1) / 2), like) @ G
        G = _S(2, 2, like) @ G @ _S(0.5, 0.5, like)
        G = _T(-0.5, -0.5, like) @ G @ _T(0.5, 0.5, like)
        Wi, Hi = 2 * (W + mx0 + mx1), 2 * (H + my0 + my1)
        Ws, Hs = (W + 6) * 2, (H + 6) * 2
        G = _S(e(2 / Wi), e(2 / Hi), like) @ G @ _S(1 / (2 / Ws), 1 / (2 / Hs), like)
        theta = G[:, :2, :].contiguous()
        flips = torch.stack([fx, fy], dim=1).to(torch.int32).contiguous()
        ws = torch.empty(hip.lib().adm_aug_workspace_floats(N, C, H, W), device=dev, dtype=torch.float32)
        out = torch.empty_like(images)
        call("adm_augment_geometric", ptr(images), ptr(flips), ptr(margin), ptr(theta), ptr(ws), ptr(out), N, C, H, W)
        return out, labels
