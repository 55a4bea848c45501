"""Autograd operators of the conditional super-resolution denoiser (BASELINE configs[4])
over the HIP C ABI: what /root/reference/unet/cond_unet_sd.py needs beyond ``adm_amd.ops``.  Same conventions: NHWC fp32 CUDA
tensors, channel counts of GEMM operands padded to multiples of 32, parameters in the reference's layouts, every backward a
hand-written HIP kernel (csrc/cond_ops.hip, csrc/attention_small.hip), no CPU / eager fallback.
"""
from __future__ import annotations

import torch

from . import hip, ops
from .hip import call, ptr
from .ops import _chk, _direct_grad, _like, _new, _notify, _Prof, ceil32, packed

_f32 = torch.float32


# ------------------------------------------------------------------------------------------------ weight standardisation
class _WeightStd(torch.autograd.Function):
    """WeightStandardizedConv2d's weight transform (cond_unet_sd.py:349-355), eps = 1e-5 (fp32)."""

    @staticmethod
    def forward(ctx, w):
        w = _chk(w, "weight")
        O, K = w.shape[0], w[0].numel()
        wn = _like(w)
        stats = _new((O, 2), w)
        call("adm_ws_fwd", ptr(w), ptr(wn), ptr(stats), O, K, 1e-5)
        ctx.save_for_backward(w, stats)
        return wn

    @staticmethod
    def backward(ctx, dwn):
        w, stats = ctx.saved_tensors
        dwn = _chk(dwn, "dwn")
        dw = _like(w)
        call("adm_ws_bwd", ptr(w), ptr(stats), ptr(dwn), ptr(dw), w.shape[0], w[0].numel(), 0)
        return dw


def weight_standardize(w):
    """Standardised copy of an OIHW weight.  Cached per parameter version when no gradient is needed (sampling)."""
    if torch.is_grad_enabled() and w.requires_grad:
        return _WeightStd.apply(w)
    key = (w.data_ptr(), w._version, ops._pack_epoch)
    ent = getattr(w, "_adm_ws", None)
    if ent is None or ent[0] != key:
        with torch.no_grad():
            ent = (key, _WeightStd.apply(w.detach()))
        try:
            w._adm_ws = ent
        except (AttributeError, TypeError):
            pass
    return ent[1]


# ------------------------------------------------------------------------------------------------ strided / large-filter convs
class _ConvGeneric(torch.autograd.Function):
    """NHWC conv with filter size ks <= 7, a stride and explicit top/left padding: the 7x7 stem (cond_unet_sd.py:653-656) and
    Downsample = Conv2d(C, C', 4, 2, 1) (:341-342).  Forward and weight gradient on the implicit-GEMM kernels (generic tap
    path); the data gradient is the transposed conv in GEMM form: col = dy x W^T (1x1 kernel), then a col2im gather."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        x = _chk(x, "x")
        B, H, W, cx = x.shape
        co, ci, ks = weight.shape[0], weight.shape[1], weight.shape[-1]
        cop, cip = ceil32(co), ceil32(ci)
        if cx != cip:
            raise RuntimeError(f"conv input has {cx} channels, expected {cip}")
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        w = _chk(weight.detach(), "weight")
        wp = _new((cop, ks * ks * cip), x)
        call("adm_pack_weight", ptr(w), ptr(wp), None, co, ci, ks, cop, cip, 0)
        bp = None
        if bias is not None:
            bp = _chk(bias.detach(), "bias")
            if cop != co:
                bp = torch.zeros((cop,), device=x.device, dtype=_f32)
                bp[:co] = bias.detach()
        y = _new((B, Ho, Wo, cop), x)
        with _Prof("igemm", 2.0 * B * Ho * Wo * co * ci * ks * ks, f"fwd-k{ks}s{stride} M={B * Ho * Wo} N={cop} K={ks * ks * cip}"):
            call("adm_conv_fwd_strided", ptr(x), ptr(wp), ptr(bp), None, ptr(y), B, H, W, Ho, Wo, cip, cip, cop, cop, cop, cop,
                 ks, stride, pad)
        ctx.save_for_backward(x, weight, bias)
        ctx.meta = (stride, pad, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        stride, pad, Ho, Wo = ctx.meta
        dy = _chk(dy, "dy")
        B, H, W, cip = x.shape
        co, ci, ks = weight.shape[0], weight.shape[1], weight.shape[-1]
        cop = ceil32(co)
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            dwp = _new((cop, ks * ks * cip), dy)
            need_b = bias is not None and ctx.needs_input_grad[2]
            dbp = torch.zeros((cop,), device=dy.device, dtype=_f32) if need_b else None
            with _Prof("wgrad", 2.0 * B * Ho * Wo * co * ci * ks * ks, f"wgrad-k{ks}s{stride} P={B * Ho * Wo} Co={cop} Ci={cip}"):
                call("adm_conv_wgrad_strided", ptr(x), ptr(dy), ptr(dwp), ptr(dbp), B, H, W, Ho, Wo, cip, cip, cop, cop, ks, stride,
                     pad)
            sink = _direct_grad(weight)
            if sink is not None:
                call("adm_unpack_wgrad", ptr(dwp), ptr(sink), co, ci, ks, cop, cip, 0, 1)
                _notify(weight)
            else:
                dw = _like(weight)
                call("adm_unpack_wgrad", ptr(dwp), ptr(dw), co, ci, ks, cop, cip, 0, 0)
            if need_b:
                bsink = _direct_grad(bias)
                if bsink is not None:
                    call("adm_add", ptr(bsink), ptr(dbp), ptr(bsink), co)
                    _notify(bias)
                else:
                    db = dbp[:co].clone()
        if ctx.needs_input_grad[0]:
            w = _chk(weight.detach(), "weight")
            wt = _new((ks * ks * cip, cop), dy)
            call("adm_pack_weight_tconv", ptr(w), ptr(wt), co, ci, ks, cop, cip)
            M = B * Ho * Wo
            col = _new((M, ks * ks * cip), dy)
            with _Prof("igemm", 2.0 * M * co * ci * ks * ks, f"dgrad-tconv M={M} N={ks * ks * cip} K={cop}"):
                call("adm_conv_fwd", ptr(dy), ptr(wt), None, None, ptr(col), 1, M, 1, cop, cop, ks * ks * cip, ks * ks * cip,
                     ks * ks * cip, ks * ks * cip, 1, 0, -1)
            dx = _new((B, H, W, cip), dy)
            call("adm_col2im", ptr(col), ptr(dx), B, H, W, Ho, Wo, cip, ks, stride, pad)
        return dx, dw, db, None, None


def conv2d_generic(x, weight, bias=None, *, stride=1, pad=0):
    return _ConvGeneric.apply(x, weight, bias, int(stride), int(pad))


# ------------------------------------------------------------------------------------------------ channel LayerNorm
class _LayerNormC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g):
        x = _chk(x, "x")
        C = x.shape[-1]
        gv = _chk(g.detach().reshape(-1), "g")
        if gv.numel() != C:
            raise RuntimeError(f"LayerNorm gain has {gv.numel()} entries, the tensor {C} channels")
        y = _like(x)
        call("adm_lnc_fwd", ptr(x), ptr(gv), ptr(y), x.numel() // C, C, 1e-5)
        ctx.save_for_backward(x, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = _chk(dy, "dy")
        C = x.shape[-1]
        M = x.numel() // C
        dx = _like(x)
        part = _new((hip.lib().adm_lnc_blocks(M) * C,), x, torch.float64)
        sink = _direct_grad(g)
        dg = sink if sink is not None else _like(g)
        call("adm_lnc_bwd", ptr(x), ptr(dy), ptr(g.detach()), ptr(dx), ptr(dg), ptr(part), M, C, 1e-5, int(sink is not None))
        if sink is not None:
            _notify(g)
            return dx, None
        return dx, dg


def layer_norm_c(x, g):
    """(x - mean_c) * rsqrt(var_c + 1e-5) * g over the channels of every pixel (cond_unet_sd.py:359-368); g is [1, C, 1, 1]."""
    return _LayerNormC.apply(x, g)


# ------------------------------------------------------------------------------------------------ BatchNorm2d
class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, training, momentum, eps):
        x = _chk(x, "x")
        C = x.shape[-1]
        M = x.numel() // C
        mr = _new((C, 2), x)
        y = _like(x)
        part = _new((hip.lib().adm_bn_blocks(M) * 2 * C,), x, torch.float64) if training else None
        call("adm_bn_fwd", ptr(x), ptr(gamma.detach()), ptr(beta.detach()), ptr(run_mean), ptr(run_var), ptr(mr), ptr(y), ptr(part),
             M, C, float(eps), float(momentum), int(training))
        ctx.save_for_backward(x, gamma, beta, mr)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mr = ctx.saved_tensors
        dy = _chk(dy, "dy")
        C = x.shape[-1]
        M = x.numel() // C
        dx = _like(x)
        part = _new((hip.lib().adm_bn_blocks(M) * 2 * C,), x, torch.float64)
        sums = _new((2 * C,), x)
        sg, sb = _direct_grad(gamma), _direct_grad(beta)
        direct = sg is not None and sb is not None
        dg = sg if direct else _like(gamma)
        db = sb if direct else _like(beta)
        call("adm_bn_bwd", ptr(x), ptr(dy), ptr(mr), ptr(gamma.detach()), ptr(dx), ptr(dg), ptr(db), ptr(part), ptr(sums), M, C,
             int(ctx.training), int(direct))
        if direct:
            _notify(gamma); _notify(beta)
            return dx, None, None, None, None, None, None, None
        return dx, dg, db, None, None, None, None, None


def batch_norm(x, bn: torch.nn.BatchNorm2d, training: bool):
    """nn.BatchNorm2d on an NHWC tensor whose channel count equals bn.num_features (a multiple of 4); in training mode the
    module's running statistics are updated in place by the kernel."""
    if x.shape[-1] != bn.num_features:
        raise RuntimeError(f"BatchNorm over {bn.num_features} channels applied to {x.shape[-1]}")
    if training:
        bn.num_batches_tracked += 1
    return _BatchNorm.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bool(training), bn.momentum, bn.eps)


# ------------------------------------------------------------------------------------------------ bilinear resize
class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, align):
        x = _chk(x, "x")
        B, Hi, Wi, C = x.shape
        y = _new((B, Ho, Wo, C), x)
        call("adm_bilinear_fwd", ptr(x), ptr(y), B, Hi, Wi, Ho, Wo, C, C, 0, int(align))
        ctx.meta = (B, Hi, Wi, Ho, Wo, C, align)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Hi, Wi, Ho, Wo, C, align = ctx.meta
        dy = _chk(dy, "dy")
        dx = _new((B, Hi, Wi, C), dy)
        call("adm_bilinear_bwd", ptr(dy), ptr(dx), B, Hi, Wi, Ho, Wo, C, C, 0, int(align))
        return dx, None, None, None


def bilinear(x, Ho: int, Wo: int, align_corners: bool):
    """F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=...) on NHWC."""
    return _Bilinear.apply(x, int(Ho), int(Wo), bool(align_corners))


def bilinear_into(x, y, coff: int, align_corners: bool):
    """No-grad: writes the resized x into channels [coff, coff + C) of the NHWC tensor y (the stem's concatenation of the
    latent with the up-sampled condition feature, cond_unet_sd.py:824)."""
    B, Hi, Wi, C = x.shape
    call("adm_bilinear_fwd", ptr(_chk(x, "x")), ptr(y), B, Hi, Wi, y.shape[1], y.shape[2], C, y.shape[3], int(coff),
         int(align_corners))
    return y


# ------------------------------------------------------------------------------------------------ activations
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, drop_p, seed):
        x = _chk(x, "x")
        if x.numel() % 4:
            raise RuntimeError("activation tensors must have a multiple of 4 elements")
        y = _like(x)
        call("adm_act_fwd", ptr(x), ptr(y), x.numel(), act, float(drop_p), int(seed))
        ctx.save_for_backward(x)
        ctx.meta = (act, drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        act, drop_p, seed = ctx.meta
        dy = _chk(dy, "dy")
        dx = _like(x)
        call("adm_act_bwd", ptr(x), ptr(dy), ptr(dx), x.numel(), act, float(drop_p), int(seed))
        return dx, None, None, None


def relu_dropout(x, drop_p: float = 0.0):
    return _Act.apply(x, 1, float(drop_p), ops.next_dropout_seed() if drop_p > 0 else 0)


def dropout(x, drop_p: float):
    return x if drop_p <= 0 else _Act.apply(x, 0, float(drop_p), ops.next_dropout_seed())


def gelu(x):
    return _Act.apply(x, 2, 0.0, 0)


def fourier_features(x, W):
    """GaussianFourierProjection (cond_unet_sd.py:396-405); W is a frozen parameter, x carries no gradient."""
    x = _chk(x.reshape(-1), "x")
    out = _new((x.shape[0], 2 * W.numel()), x)
    call("adm_fourier_features", ptr(x), ptr(_chk(W.detach(), "W")), ptr(out), x.shape[0], W.numel())
    return out


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _chk(a, "a"), _chk(b, "b")
        if a.shape != b.shape:
            raise RuntimeError(f"add: shapes {tuple(a.shape)} and {tuple(b.shape)} differ")
        y = _like(a)
        call("adm_add", ptr(a), ptr(b), ptr(y), a.numel())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


def avg_pool(x, win: int):
    """AvgPool2d(kernel = stride = win) for power-of-two windows that divide the map (every window of the DIV2K recipe at its
    training sizes): repeated 2x2 means (means of equal-size groups compose exactly)."""
    if win & (win - 1):
        raise NotImplementedError("window sizes must be powers of two")
    if x.shape[1] % win or x.shape[2] % win:
        raise NotImplementedError(f"AvgPool window {win} does not divide the {x.shape[1]}x{x.shape[2]} map (the reference "
                                  "zero-pads in that case; not implemented)")
    while win > 1:
        x = ops.downsample2x(x)
        win >>= 1
    return x


# ------------------------------------------------------------------------------------------------ attention
class _MHA(torch.autograd.Function):
    """softmax(scale q k^T) v with separate query / key lengths; q [B, Lq, H*D], k, v [B, Lk, H*D] (contiguous)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        q, k, v = _chk(q, "q"), _chk(k, "k"), _chk(v, "v")
        B, Lq, C = q.shape
        Lk = k.shape[1]
        D = C // heads
        o = _like(q)
        lse = _new((B * heads, Lq), q)
        call("adm_mha_fwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), B, Lq, Lk, heads, D, C, C, C, C, float(scale))
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.meta = (heads, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        do = _chk(do, "do")
        B, Lq, C = q.shape
        Lk = k.shape[1]
        dq, dk, dv = _like(q), _like(k), _like(v)
        delta = _like(lse)
        call("adm_mha_bwd", ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(dq), ptr(dk), ptr(dv), ptr(delta), B, Lq, Lk,
             heads, C // heads, C, C, C, C, C, C, C, float(scale))
        return dq, dk, dv, None, None


def mha(q, k, v, heads: int, scale: float = 1.0):
    return _MHA.apply(q, k, v, int(heads), float(scale))


class _SelfAttnPacked(torch.autograd.Function):
    """Attention over a packed qkv tensor [B, L, 3*H*D] = (q | k | v), each (head, channel) (cond_unet_sd.py:544-553)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        qkv = _chk(qkv, "qkv")
        B, L, C3 = qkv.shape
        C = C3 // 3
        o = _new((B, L, C), qkv)
        lse = _new((B * heads, L), qkv)
        base = qkv.data_ptr()
        P = hip.c_void_p
        call("adm_mha_fwd", P(base), P(base + 4 * C), P(base + 8 * C), ptr(o), ptr(lse), B, L, L, heads, C // heads, C3, C3, C3, C,
             float(scale))
        ctx.save_for_backward(qkv, o, lse)
        ctx.meta = (heads, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        do = _chk(do, "do")
        B, L, C3 = qkv.shape
        C = C3 // 3
        dqkv = _like(qkv)
        delta = _like(lse)
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        P = hip.c_void_p
        call("adm_mha_bwd", P(base), P(base + 4 * C), P(base + 8 * C), ptr(o), ptr(do), ptr(lse), P(dbase), P(dbase + 4 * C),
             P(dbase + 8 * C), ptr(delta), B, L, L, heads, C // heads, C3, C3, C3, C, C3, C3, C3, float(scale))
        return dqkv, None, None


def self_attention_packed(qkv, heads: int, scale: float):
    return _SelfAttnPacked.apply(qkv, int(heads), float(scale))


class _LinearAttention(torch.autograd.Function):
    """LinearAttention core on qkv [B, N, 384] (4 heads x 32) -> [B, N, 128] (cond_unet_sd.py:516-529)."""

    @staticmethod
    def forward(ctx, qkv):
        qkv = _chk(qkv, "qkv")
        B, N, C3 = qkv.shape
        if C3 != 384:
            raise RuntimeError("LinearAttention is specialised to 4 heads x 32 channels (to_qkv with 384 outputs)")
        out = _new((B, N, 128), qkv)
        cx = _new((B, 4, 32, 32), qkv)
        kst = _new((B, 128, 2), qkv)
        ws = _new((hip.lib().adm_linattn_ws_floats(B, N),), qkv)
        call("adm_linattn_fwd", ptr(qkv), ptr(out), ptr(cx), ptr(kst), ptr(ws), B, N)
        ctx.save_for_backward(qkv, cx, kst)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, cx, kst = ctx.saved_tensors
        dout = _chk(dout, "dout")
        B, N, _ = qkv.shape
        dqkv = _like(qkv)
        dctx = _like(cx)
        S = _new((B, 128), qkv)
        ws = _new((hip.lib().adm_linattn_ws_floats(B, N),), qkv)
        call("adm_linattn_bwd", ptr(qkv), ptr(dout), ptr(cx), ptr(kst), ptr(dqkv), ptr(dctx), ptr(S), ptr(ws), B, N)
        return dqkv


def linear_attention(qkv):
    return _LinearAttention.apply(qkv)


class _SpatialAttBig(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att, qk, h, xres):
        att, qk, h, xres = _chk(att, "att"), _chk(qk, "qk"), _chk(h, "h"), _chk(xres, "xres")
        B, H, W, C = h.shape
        y = _like(h)
        gate = _new((B, H * W, 2), h)
        call("adm_spatial_att_big_fwd", ptr(att), att.shape[-1], ptr(qk), ptr(h), ptr(xres), ptr(y), ptr(gate), B, H * W, C)
        ctx.save_for_backward(att, qk, h, gate)
        return y

    @staticmethod
    def backward(ctx, dy):
        att, qk, h, gate = ctx.saved_tensors
        dy = _chk(dy, "dy")
        B, H, W, C = h.shape
        dh, datt = _like(h), _like(att)
        dqk = torch.zeros_like(qk)
        part = _new((B, 4), h)
        call("adm_spatial_att_big_bwd", ptr(att), att.shape[-1], ptr(qk), ptr(h), ptr(dy), ptr(gate), ptr(dh), ptr(datt), ptr(dqk),
             ptr(part), B, H * W, C)
        return datt, dqk, dh, dy


def spatial_att_gate(att, qk, h, xres):
    """ops.spatial_att_gate for any map size: the <= 64-pixel kernel of the unconditional UNet's 4x4 bottleneck, else the
    recomputing kernel (the SR denoiser's 16x16 bottleneck)."""
    if h.shape[1] * h.shape[2] <= 64:
        return ops.spatial_att_gate(att, qk, h, xres)
    return _SpatialAttBig.apply(att, qk, h, xres)
