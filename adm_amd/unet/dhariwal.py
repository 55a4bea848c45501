"""MI355X-native two-output Dhariwal UNet + DDM preconditioning behind the reference's module API.

Drop-in for /root/reference/unet/uncond_unet{,_2,_sd,_sd_2,_sd_3}.py: same constructor keywords,
same ``forward(x, sigma, class_labels=None, force_fp32=False, **kw) -> (D_x, D_y)``, same
state_dict names / shapes (OIHW weights, `resample_filter` buffers), so reference checkpoints load.
Internally everything runs NHWC through the HIP kernels in ``adm_amd.ops`` -- there is no PyTorch
fallback; on a CPU tensor the ops raise.

Reference lines restated: layers :53-129, UNetBlock :157-211, PositionalEmbedding :217-230,
DhariwalUNet :450-581, EDMPrecond :588-638, SpatialAtt :19-37, weight_init :42-47.
"""
from __future__ import annotations

import math
from typing import Optional

import weakref

import torch
import torch.nn as nn

from .. import ops

VARIANTS = ("uncond_unet", "uncond_unet_2", "uncond_unet_sd", "uncond_unet_sd_2", "uncond_unet_sd_3")


def _init_tensor(shape, mode: str, fan_in: int, fan_out: int) -> torch.Tensor:
    """The four init modes of weight_init (uncond_unet.py:42-47)."""
    if mode == "xavier_uniform":
        return math.sqrt(6 / (fan_in + fan_out)) * (torch.rand(*shape) * 2 - 1)
    if mode == "xavier_normal":
        return math.sqrt(2 / (fan_in + fan_out)) * torch.randn(*shape)
    if mode == "kaiming_uniform":
        return math.sqrt(3 / fan_in) * (torch.rand(*shape) * 2 - 1)
    if mode == "kaiming_normal":
        return math.sqrt(1 / fan_in) * torch.randn(*shape)
    raise ValueError(f'Invalid init mode "{mode}"')


def _pad_cols(x: torch.Tensor, n: int) -> torch.Tensor:
    if x.shape[-1] == n:
        return x
    y = torch.zeros(*x.shape[:-1], n, device=x.device, dtype=x.dtype)
    y[..., : x.shape[-1]] = x
    return y


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True, init_mode="kaiming_normal", init_weight=1, init_bias=0):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        kw = dict(mode=init_mode, fan_in=in_features, fan_out=out_features)
        self.weight = nn.Parameter(_init_tensor([out_features, in_features], **kw) * init_weight)
        self.bias = nn.Parameter(_init_tensor([out_features], **kw) * init_bias) if bias else None

    def forward(self, x, residual=None):
        return ops.linear(_pad_cols(x, ops.ceil32(self.in_features)), self.weight, self.bias, residual)


class Conv2d(nn.Module):
    """3x3 / 1x1 / weight-less resampling layer on NHWC tensors (reference: uncond_unet.py:72-113
    with resample_filter=[1,1], fused_resample=False)."""

    def __init__(self, in_channels, out_channels, kernel, bias=True, up=False, down=False, resample_filter=(1, 1),
                 fused_resample=False, init_mode="kaiming_normal", init_weight=1, init_bias=0, qkv=False):
        assert not (up and down)
        super().__init__()
        if list(resample_filter) != [1, 1] or fused_resample:
            raise NotImplementedError("only the box resample filter [1,1] (all DDM configs) is implemented")
        self.in_channels, self.out_channels, self.up, self.down, self.qkv = in_channels, out_channels, up, down, qkv
        kw = dict(mode=init_mode, fan_in=in_channels * kernel * kernel, fan_out=out_channels * kernel * kernel)
        self.weight = nn.Parameter(_init_tensor([out_channels, in_channels, kernel, kernel], **kw) * init_weight) if kernel else None
        self.bias = nn.Parameter(_init_tensor([out_channels], **kw) * init_bias) if kernel and bias else None
        self.register_buffer("resample_filter", torch.full((1, 1, 2, 2), 0.25) if (up or down) else None)

    def forward(self, x, residual=None):
        if self.down:
            x = ops.downsample2x(x)
        if self.weight is None:
            if self.up:
                x = ops.upsample2x(x)
            assert residual is None
            return x
        return ops.conv2d(x, self.weight, self.bias, residual, up=self.up, qkv=self.qkv)


class GroupNorm(nn.Module):
    def __init__(self, num_channels, num_groups=32, min_channels_per_group=4, eps=1e-5):
        super().__init__()
        self.num_groups = min(num_groups, num_channels // min_channels_per_group)
        if self.num_groups != min(32, num_channels // 4) or eps != 1e-5:
            raise NotImplementedError("GroupNorm kernels are specialised to groups=min(32, C//4), eps=1e-5")
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def forward(self, x, scale_shift=None, silu=False, drop_p=0.0, to_conv=False):
        """to_conv: the result feeds ops.conv2d and nothing else (lets the bf16 mode store it as bf16, ops.group_norm_act)."""
        seed = ops.next_dropout_seed() if drop_p > 0 else 0
        return ops.group_norm_act(x, self.weight, self.bias, scale_shift, silu=silu, drop_p=drop_p, seed=seed, to_conv=to_conv)

    def fork(self, x, silu=False, to_conv=False, bound=False):
        """(norm(x), x): x comes back for the residual branch so that both gradients are summed inside the GroupNorm
        backward kernel (ops.group_norm_act_fork).  bound: also leave max |norm(x)| for a conv further down (ops.group_norm_act)."""
        return ops.group_norm_act_fork(x, self.weight, self.bias, None, silu=silu, to_conv=to_conv, bound=bound)


_AFFINE_GROUPS = weakref.WeakKeyDictionary()      # DhariwalUNet -> (ops.AffineGroup over its blocks' `affine` Linears, the blocks)


class UNetBlock(nn.Module):
    def __init__(self, in_channels, out_channels, emb_channels, up=False, down=False, attention=False, num_heads=None,
                 channels_per_head=64, dropout=0, skip_scale=1, eps=1e-5, resample_filter=(1, 1), resample_proj=False,
                 adaptive_scale=True, init=dict(), init_zero=dict(init_weight=0), init_attn=None):
        super().__init__()
        if not adaptive_scale or skip_scale != 1:
            raise NotImplementedError("only the Dhariwal flavour (adaptive_scale=True, skip_scale=1) is implemented")
        self.in_channels, self.out_channels, self.emb_channels = in_channels, out_channels, emb_channels
        self.num_heads = 0 if not attention else num_heads if num_heads is not None else out_channels // channels_per_head
        if self.num_heads and out_channels != 64 * self.num_heads:
            raise NotImplementedError("attention kernels are specialised to head dim 64")
        self.dropout, self.skip_scale, self.adaptive_scale = dropout, skip_scale, adaptive_scale
        self.norm0 = GroupNorm(in_channels, eps=eps)
        self.conv0 = Conv2d(in_channels, out_channels, 3, up=up, down=down, resample_filter=resample_filter, **init)
        self.affine = Linear(emb_channels, out_channels * 2, **init)
        self.norm1 = GroupNorm(out_channels, eps=eps)
        self.conv1 = Conv2d(out_channels, out_channels, 3, **init_zero)
        self.skip = None
        if out_channels != in_channels or up or down:
            kernel = 1 if resample_proj or out_channels != in_channels else 0
            self.skip = Conv2d(in_channels, out_channels, kernel, up=up, down=down, resample_filter=resample_filter, **init)
        if self.num_heads:
            self.norm2 = GroupNorm(out_channels, eps=eps)
            self.qkv = Conv2d(out_channels, out_channels * 3, 1, qkv=True, **(init_attn if init_attn is not None else init))
            self.proj = Conv2d(out_channels, out_channels, 1, **init_zero)

    def forward(self, x, emb, ss=None):
        """ss: this block's scale/shift when the caller computed all blocks' `affine` Linears as one GEMM (ops.affine_group)."""
        # (to_conv: these normalised tensors go straight into a conv -- conv0 resamples first when it down-samples)
        n0, x = self.norm0.fork(x, silu=True, to_conv=not self.conv0.down, bound=True)    # x feeds the normalised branch AND the residual / skip branch
        h = self.conv0(n0)
        p = self.dropout if self.training else 0.0
        h = self.norm1(h, self.affine(emb) if ss is None else ss, silu=True, drop_p=p, to_conv=True)
        h = self.conv1(h, residual=x if self.skip is None else self.skip(x))
        if self.num_heads:
            n2, h = self.norm2.fork(h, to_conv=True)
            a = ops.attention(self.qkv(n2), self.num_heads)
            h = self.proj(a, residual=h)
        return h


class PositionalEmbedding(nn.Module):
    def __init__(self, num_channels, max_positions=10000, endpoint=False):
        super().__init__()
        if max_positions != 10000 or endpoint:
            raise NotImplementedError
        self.num_channels = num_channels

    def forward(self, x):
        return ops.pos_embedding(x, self.num_channels)


class SpatialAtt(nn.Module):
    """Parameter holder for decouple{1,2}[1]; the arithmetic is fused in ops.spatial_att_gate."""

    def __init__(self, in_dim):
        super().__init__()
        self.map = nn.Conv2d(in_dim, 1, 1)
        self.q_conv = nn.Conv2d(1, 1, 1)
        self.k_conv = nn.Conv2d(1, 1, 1)


def _decouple(seq: nn.Sequential, x):
    """decouple(x) + x  (uncond_unet.py:500-507, 566-567)."""
    conv, sa = seq[0], seq[1]
    x, xres = ops.fanout(x, 2)         # the conv branch and the residual
    h = ops.conv2d(x, conv.weight, conv.bias)
    qk = torch.cat([sa.q_conv.weight.reshape(1), sa.q_conv.bias, sa.k_conv.weight.reshape(1), sa.k_conv.bias])
    from .. import ops_cond            # maps larger than 8x8 (a bottleneck above 4x4) take the recomputing kernel
    h, hg = ops.fanout(h, 2)           # ... and h feeds the attention map and the gate
    att = ops.conv2d(h, sa.map.weight, sa.map.bias)
    return ops_cond.spatial_att_gate(att, qk, hg, xres)


class DhariwalUNet(nn.Module):
    def __init__(self, img_resolution, in_channels, out_channels, label_dim=0, augment_dim=0, model_channels=192,
                 channel_mult=(1, 2, 3, 4), channel_mult_emb=4, num_blocks=3, attn_resolutions=(32, 16, 8), dropout=0.10,
                 label_dropout=0, out_mul=1, variant="uncond_unet", **kwargs):
        super().__init__()
        if label_dim:
            raise NotImplementedError("class-conditional labels are not on the DDM hot path")
        if out_mul != 1:
            raise NotImplementedError("out_mul != 1 (ddm_linear) is out of scope")
        if any((model_channels * m) % 32 for m in channel_mult):
            raise NotImplementedError("channel widths must be multiples of 32 (the HIP GEMM operands' channel granularity); "
                                      f"got model_channels={model_channels}, channel_mult={list(channel_mult)}")
        self.variant = variant
        self.two_decoders = variant in ("uncond_unet", "uncond_unet_2")
        self.label_dropout = label_dropout
        self.in_channels_pad = ops.ceil32(in_channels)
        emb_channels = model_channels * channel_mult_emb
        init = dict(init_mode="kaiming_uniform", init_weight=math.sqrt(1 / 3), init_bias=math.sqrt(1 / 3))
        init_zero = dict(init_mode="kaiming_uniform", init_weight=0, init_bias=0)
        init_one = dict(init_mode="kaiming_uniform", init_weight=1, init_bias=0)
        bk = dict(emb_channels=emb_channels, channels_per_head=64, dropout=dropout, init=init, init_zero=init_zero)

        self.map_noise = PositionalEmbedding(model_channels)
        self.map_augment = Linear(augment_dim, model_channels, bias=False, **init_zero) if augment_dim else None
        self.map_layer0 = Linear(model_channels, emb_channels, **init)
        self.map_layer1 = Linear(emb_channels, emb_channels, **init)
        self.map_label = None

        self.enc = nn.ModuleDict()
        cout = in_channels
        for level, mult in enumerate(channel_mult):
            res = img_resolution >> level
            if level == 0:
                cin, cout = cout, model_channels * mult
                self.enc[f"{res}x{res}_conv"] = Conv2d(cin, cout, 3, **init)
            else:
                self.enc[f"{res}x{res}_down"] = UNetBlock(cout, cout, down=True, **bk)
            for idx in range(num_blocks):
                cin, cout = cout, model_channels * mult
                self.enc[f"{res}x{res}_block{idx}"] = UNetBlock(cin, cout, attention=(res in attn_resolutions), **bk)
        skips = [b.out_channels for b in self.enc.values()]

        def make_decouple(c):
            return nn.Sequential(nn.Conv2d(c, c, 3, 1, 1), SpatialAtt(c))

        def make_decoder(cout, skips):
            dec = nn.ModuleDict()
            for level, mult in reversed(list(enumerate(channel_mult))):
                res = img_resolution >> level
                if level == len(channel_mult) - 1:
                    dec[f"{res}x{res}_in0"] = UNetBlock(cout, cout, attention=True, **bk)
                    dec[f"{res}x{res}_in1"] = UNetBlock(cout, cout, **bk)
                else:
                    dec[f"{res}x{res}_up"] = UNetBlock(cout, cout, up=True, **bk)
                for idx in range(num_blocks + 1):
                    cin = cout + skips.pop()
                    cout = model_channels * mult
                    dec[f"{res}x{res}_block{idx}"] = UNetBlock(cin, cout, attention=(res in attn_resolutions), **bk)
            return dec, cout

        self.decouple1 = make_decouple(cout)
        if self.two_decoders:
            self.decouple2 = make_decouple(cout)
        self.dec, c1 = make_decoder(cout, list(skips))
        self.out_norm = GroupNorm(c1)
        self.out_conv = Conv2d(c1, out_channels * out_mul, 3, **init_one)
        if self.two_decoders:
            self.dec2, c2 = make_decoder(cout, list(skips))
            self.out_norm2 = GroupNorm(c2)
            self.out_conv2 = Conv2d(c2, out_channels, 3, **init_one)

    def embed(self, noise_labels, augment_labels=None):
        emb = self.map_noise(noise_labels)
        if self.map_augment is not None and augment_labels is not None:
            emb = self.map_augment(augment_labels.to(torch.float32), residual=emb)
        emb = ops.silu(self.map_layer0(emb))
        return ops.silu(self.map_layer1(emb))

    def _decode(self, dec, x, skips, emb, out_norm, out_conv, ratios=None, ss=None):
        stack = list(skips)
        ratios = list(ratios) if ratios is not None else None
        for block in dec.values():
            if x.shape[-1] != block.in_channels:
                r = ratios.pop() if ratios is not None else 1.0
                x = ops.concat_channels(x, stack.pop(), r)
            x = block(x, emb, None if ss is None else ss[block])
        return out_conv(out_norm(x, silu=True, to_conv=True))

    def _scale_shifts(self, emb):
        """{block: its scale/shift} from ONE GEMM over the `affine` Linears of every block (ops.affine_group), or None."""
        if not ops.AFFINE_GROUP or ops.DETERMINISTIC or ops.COMPUTE != "f32" or not emb.is_cuda:
            return None
        grp = _AFFINE_GROUPS.get(self)          # (kept outside the module: deepcopy / state_dict / pickling see nothing of it)
        if grp is None:
            blocks = [b for b in self.modules() if isinstance(b, UNetBlock)]
            grp = _AFFINE_GROUPS[self] = (ops.AffineGroup([b.affine for b in blocks]), blocks)
        group, blocks = grp
        return dict(zip(blocks, ops.affine_group(emb, group)))

    def forward(self, x, noise_labels, class_labels=None, augment_labels=None, **kwargs):
        """x: NHWC [B,H,W,32] (3 real channels).  Returns (F_x, F_y) NHWC with 32 padded channels
        (F_y is None for single-decoder variants)."""
        emb = self.embed(noise_labels, augment_labels)
        ss = self._scale_shifts(emb)
        skips, skips2 = [], []
        for block in self.enc.values():
            x = block(x, emb, None if ss is None else ss[block]) if isinstance(block, UNetBlock) else block(x)
            # an encoder output feeds the next block and one concatenation per decoder: its gradient is their sum (one launch)
            x, sk1, sk2 = ops.fanout(x, 3) if self.two_decoders else (*ops.fanout(x, 2), None)
            skips.append(sk1); skips2.append(sk2)
        ratios = None
        if self.variant == "uncond_unet_sd_3":          # skip-tuning (uncond_unet_sd_3.py:547-555)
            n = len(skips)
            ratios = [0.5 + 0.5 * i / (n - 1) for i in range(n)][::-1]
        f_y = None
        x, x2 = ops.fanout(x, 2) if self.two_decoders else (x, None)      # the bottleneck feeds both decouple modules
        s2 = ops.branch_stream() if (self.two_decoders and x.is_cuda) else None
        if s2 is not None:      # the second decoder on its own stream, concurrently with the first (ops.BRANCH_STREAM)
            main = torch.cuda.current_stream()
            with torch.cuda.stream(s2):
                f_y = self._decode(self.dec2, _decouple(self.decouple2, x2), skips2, emb, self.out_norm2, self.out_conv2, ss=ss)
            f_x = self._decode(self.dec, _decouple(self.decouple1, x), skips, emb, self.out_norm, self.out_conv, ratios, ss=ss)
            main.wait_stream(s2)
            return f_x, f_y
        f_x = self._decode(self.dec, _decouple(self.decouple1, x), skips, emb, self.out_norm, self.out_conv, ratios, ss=ss)
        if self.two_decoders:
            f_y = self._decode(self.dec2, _decouple(self.decouple2, x2), skips2, emb, self.out_norm2, self.out_conv2, ss=ss)
        return f_x, f_y


class EDMPrecond(nn.Module):
    """DDM preconditioning around DhariwalUNet; subclasses in uncond_unet*.py pin ``VARIANT``."""

    VARIANT = "uncond_unet"

    def __init__(self, img_resolution, img_channels, label_dim=0, use_fp16=False, sigma_min=0, sigma_max=float("inf"),
                 sigma_data=0.5, model_type="DhariwalUNet", precondition=True, **model_kwargs):
        super().__init__()
        if model_type != "DhariwalUNet":
            raise NotImplementedError(f"model_type {model_type!r}: every DDM config selects DhariwalUNet")
        if use_fp16:
            raise NotImplementedError("fp16 execution is not implemented (all reference configs run fp32)")
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.self_condition = None
        self.precondition = precondition
        self.channels = img_channels
        self.label_dim, self.use_fp16 = label_dim, use_fp16
        self.sigma_min, self.sigma_max, self.sigma_data = sigma_min, sigma_max, sigma_data
        model_kwargs.pop("cfg", None)                 # sample_uncond.py:47-49 passes cfg= through
        self.model = DhariwalUNet(img_resolution=img_resolution, in_channels=img_channels, out_channels=img_channels,
                                  label_dim=label_dim, variant=self.VARIANT, **model_kwargs)

    def coefficients(self, sigma: torch.Tensor):
        """(c_skip1, c_out1, c_skip2, c_out2, c_in, c_noise), each [B] fp32 (B may be 1)."""
        if self.VARIANT in ("uncond_unet", "uncond_unet_sd"):        # uncond_unet.py:621-626
            den = sigma ** 2 - sigma + 1
            c_skip1 = (sigma - 1) / den
            c_skip2 = sigma.sqrt() / den
            c_out1 = torch.sqrt(sigma / den)
            c_out2 = (1 - sigma) / den.sqrt()
            c_in = 1 / torch.sqrt((1 - sigma) ** 2 + sigma)
        else:                                                         # uncond_unet_2.py:623-627
            den = sigma ** 2 + (sigma - 1) ** 2
            c_skip1 = (sigma - 1) / den
            c_out1 = sigma / den.sqrt()
            c_skip2 = sigma / den
            c_out2 = (1 - sigma) / den.sqrt()
            c_in = 1 / den.sqrt()
        return c_skip1, c_out1, c_skip2, c_out2, c_in, sigma.log()

    def forward(self, x, sigma, class_labels=None, force_fp32=False, *args, **model_kwargs):
        dev = x.device
        sigma = torch.as_tensor(sigma, device=dev).to(torch.float32).reshape(-1)
        c_skip1, c_out1, c_skip2, c_out2, c_in, c_noise = [c.contiguous() for c in self.coefficients(sigma)]
        aug = model_kwargs.get("augment_labels", None)
        xin = ops.nchw_to_nhwc(x, c_in, self.model.in_channels_pad)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        x = x.contiguous()
        f_x, f_y = self.model(xin, c_noise, None, augment_labels=aug)
        one, zero = torch.ones_like(sigma), torch.zeros_like(sigma)
        if self.precondition:
            d_x = ops.precond_out(f_x, x, c_skip1, c_out1)
        else:
            d_x = ops.precond_out(f_x, x, zero, one)
        if self.model.two_decoders:
            d_y = ops.precond_out(f_y, x, c_skip2, c_out2) if self.precondition else ops.precond_out(f_y, x, zero, one)
        else:   # D_y = (x - (sigma-1) D_x) / g(sigma)   (uncond_unet_sd.py:602, uncond_unet_sd_2.py:603)
            g = sigma.sqrt() if self.VARIANT == "uncond_unet_sd" else sigma
            d_y = ops.axpby_batch(d_x, x, (1 / g).contiguous(), (-(sigma - 1) / g).contiguous())
        return d_x, d_y

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)
