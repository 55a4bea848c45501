"""KL autoencoder ("first stage", KL-f4 for the CelebA-HQ latent configs) on the HIP hot path -- inference only.

Mirror of the part of /root/reference/ddm/encoder_decoder.py that ``ddm_const_2.LatentDiffusion`` uses:
``AutoencoderKL(ddconfig, lossconfig, embed_dim, ckpt_path=...)`` with ``.encode(x) -> DiagonalGaussianDistribution``,
``.decode(z)``, ``.down_ratio`` (encoder_decoder.py:894-946); Encoder :386-479, Decoder :482-587, ResnetBlock :99-160,
AttnBlock :169-213, Downsample :78-96, Upsample :60-75, DiagonalGaussianDistribution :854-892.

The submodules are ordinary ``torch.nn.Conv2d`` / ``torch.nn.GroupNorm`` objects used ONLY as parameter containers
(same names, shapes and default initialisation as the reference, so its checkpoints load with ``strict`` key
matching, minus the ``loss.*`` LPIPS/discriminator entries that only AE *training* needs); their ``forward`` is never
called.  All arithmetic runs through adm_amd.ops on NHWC fp32 buffers:
  3x3 / 1x1 convs        adm_conv_fwd (fp32 MFMA implicit GEMM), Upsample fused as the nearest-x2 loader mode,
                         Downsample = adm_conv_fwd_strided (stride 2, zero padding bottom/right only)
  Normalize + swish      adm_gn_stats / adm_gn_apply with 32 groups, eps 1e-6
  mid-block attention    single head, d = C: QK^T and PV on the same implicit-GEMM kernel per image
                         (scores [L, L] live in HBM: 64 MB at L = 4096), adm_softmax_rows in between; V is produced
                         already transposed ([C, L] = W_v h^T) and its bias added after PV (softmax rows sum to 1)
  posterior sample       adm_posterior_sample
The first stage is frozen (ddm_const_2.py:436-440): everything here is forward-only and must run under no_grad.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import ops

_CHUNK_BYTES = 1 << 30     # activations of one pass stay below 1 GiB (32-bit buffer offsets need < 2 GiB)


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _gn(norm: nn.GroupNorm, x, silu: bool):
    return ops.group_norm_act(x, norm.weight, norm.bias, silu=silu, groups=norm.num_groups, eps=norm.eps)


def _conv(conv: nn.Conv2d, x, residual=None, up=False):
    return ops.conv2d(x, conv.weight, conv.bias, residual, up=up)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=False is not used by any DDM config")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        return _conv(self.conv, x, up=True)          # nearest x2 fused into the conv's loader


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=False is not used by any DDM config")
        self.with_conv = with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x):
        return ops.conv2d_strided(x, self.conv.weight, self.conv.bias, stride=2, pad_lo=0, pad_hi=1)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if conv_shortcut or temb_channels > 0:
            raise NotImplementedError("conv_shortcut / timestep embedding are not used by the autoencoder")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)            # inactive: the first stage always runs in eval mode
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, temb=None):
        h = _conv(self.conv1, _gn(self.norm1, x, True))
        h = _gn(self.norm2, h, True)
        if self.in_channels != self.out_channels:
            x = _conv(self.nin_shortcut, x)
        return _conv(self.conv2, h, residual=x)       # x + h fused into the conv epilogue


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def forward(self, x):
        B, H, W, C = x.shape
        L = H * W
        if C % 32 or L % 32:
            raise RuntimeError(f"AttnBlock needs C ({C}) and H*W ({L}) to be multiples of 32")
        h = _gn(self.norm, x, False)
        q = _conv(self.q, h).reshape(B, L, C)
        k = _conv(self.k, h).reshape(B, L, C)
        hf = h.reshape(B, L, C)
        wv = self.v.weight.detach().reshape(C, C)
        o = torch.empty_like(q)
        s = torch.empty((L, L), device=x.device, dtype=torch.float32)
        vt = torch.empty((C, L), device=x.device, dtype=torch.float32)
        for b in range(B):
            ops.matmul_nt(q[b], k[b], out=s)                      # s[i][j] = q_i . k_j
            ops.softmax_rows_(s, float(C) ** -0.5)
            ops.matmul_nt(wv, hf[b], out=vt)                      # V^T (bias deferred)
            ops.matmul_nt(s, vt, bias=self.v.bias.detach(), out=o[b])
        return _conv(self.proj_out, o.reshape(B, H, W, C), residual=x)


def make_attn(in_channels, attn_type="vanilla"):
    if attn_type == "vanilla":
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity(in_channels)
    raise NotImplementedError(f"attn_type {attn_type!r} (linear attention is not used by any DDM config)")


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        if use_linear_attn:
            attn_type = "linear"
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = (curr_res[0] // 2, curr_res[1] // 2)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        """x: NHWC fp32 [B, H, W, ceil32(in_channels)] -> NHWC [B, H/f, W/f, ceil32(2 z)]."""
        h = _conv(self.conv_in, x)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block](h)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block](h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample(h)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        return _conv(self.conv_out, _gn(self.norm_out, h, True))


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if use_linear_attn:
            attn_type = "linear"
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are not used by any DDM config")
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = (resolution[0] // 2 ** (self.num_resolutions - 1), resolution[1] // 2 ** (self.num_resolutions - 1))
        self.z_shape = (1, z_channels, curr_res[0], curr_res[1])
        self.conv_in = nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = (curr_res[0] * 2, curr_res[1] * 2)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z):
        """z: NHWC fp32 [B, h, w, ceil32(z_channels)] -> NHWC [B, h f, w f, ceil32(out_ch)]."""
        h = _conv(self.conv_in, z)
        h = self.mid.block_1(h)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block](h)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block](h)
            if i_level != 0:
                h = self.up[i_level].upsample(h)
        return _conv(self.conv_out, _gn(self.norm_out, h, True))


class DiagonalGaussianDistribution(object):
    """Posterior over latents.  Holds the NHWC moments the encoder produced; ``sample`` / ``mode`` run the HIP kernel
    and return NCHW like the reference.  ``parameters`` / ``mean`` / ``logvar`` / ``std`` / ``var`` are NCHW views for
    API parity (encoder_decoder.py:854-892)."""

    def __init__(self, moments_nhwc, channels: int, deterministic=False):
        self._m = moments_nhwc
        self._c = channels
        self.deterministic = deterministic

    @property
    def parameters(self):
        return self._m[..., :2 * self._c].permute(0, 3, 1, 2).contiguous()

    @property
    def mean(self):
        return self._m[..., :self._c].permute(0, 3, 1, 2).contiguous()

    @property
    def logvar(self):
        return torch.clamp(self._m[..., self._c:2 * self._c].permute(0, 3, 1, 2), -30.0, 20.0).contiguous()

    @property
    def std(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(self.logvar)

    def sample(self, eps: Optional[torch.Tensor] = None):
        """mean + std * N(0,1); ``eps`` ([B,C,h,w]) injects the draw for parity tests."""
        if self.deterministic:
            return self.mode()
        B, H, W, _ = self._m.shape
        if eps is None:
            e = torch.randn((B, H, W, self._c), device=self._m.device, dtype=torch.float32)
        else:
            e = eps.to(device=self._m.device, dtype=torch.float32).permute(0, 2, 3, 1).contiguous()
        return ops.posterior_sample(self._m, self._c, e).permute(0, 3, 1, 2).contiguous()

    def mode(self):
        return ops.posterior_sample(self._m, self._c, None).permute(0, 3, 1, 2).contiguous()

    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0
                               - self.logvar + other.logvar, dim=[1, 2, 3])


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=3, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, **kwargs):
        super().__init__()
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.down_ratio = 2 ** (len(ddconfig["ch_mult"]) - 1)
        # `lossconfig` (LPIPSWithDiscriminator) only matters for autoencoder TRAINING, which is out of scope; its
        # `loss.*` checkpoint entries are skipped on load.
        assert ddconfig["double_z"]
        self.z_channels = ddconfig["z_channels"]
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        if colorize_nlabels is not None:
            self.register_buffer("colorize", torch.randn(3, colorize_nlabels, 1, 1))
        if monitor is not None:
            self.monitor = monitor
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=(), use_ema=True):
        """Checkpoint layouts of encoder_decoder.py:918-935 ('ema' / 'model' / 'state_dict'); tensors only."""
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if "ema" in sd and use_ema:
            sd = {k[10:]: v for k, v in sd["ema"].items() if k.startswith("ema_model.")}
        elif "model" in sd:
            sd = sd["model"]
        elif "state_dict" in sd:
            sd = sd["state_dict"]
        else:
            raise ValueError("checkpoint has none of 'ema', 'model', 'state_dict'")
        for k in list(sd.keys()):
            if k.startswith("loss.") or any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        msg = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")
        print("==>Load AutoEncoder Info: ", msg)

    # ---------------------------------------------------------------------------------------------
    def _chunks(self, B, H, W):
        widest = max(self.encoder.ch * max(self.encoder.in_ch_mult), 32)
        per_img = H * W * widest * 4
        n = max(1, min(B, _CHUNK_BYTES // max(per_img, 1)))
        return [(i, min(B, i + n)) for i in range(0, B, n)]

    @torch.no_grad()
    def encode(self, x):
        """x NCHW in [-1, 1] -> posterior (encoder_decoder.py:937-941)."""
        B, C, H, W = x.shape
        outs = []
        with ops.batch_invariant(B):        # every image gets the same bits whatever the chunking (kernel choice by the whole batch)
            for lo, hi in self._chunks(B, H, W):
                xh = ops.nchw_to_nhwc(x[lo:hi], None, ops.ceil32(C))
                outs.append(_conv(self.quant_conv, self.encoder(xh)))
        moments = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return DiagonalGaussianDistribution(moments, self.embed_dim)

    @torch.no_grad()
    def decode(self, z):
        """z NCHW [B, embed_dim, h, w] -> image NCHW [B, out_ch, h f, w f] (encoder_decoder.py:943-946)."""
        B, C, h, w = z.shape
        f = self.down_ratio
        outs = []
        with ops.batch_invariant(B):
            for lo, hi in self._chunks(B, h * f, w * f):
                zh = ops.nchw_to_nhwc(z[lo:hi].to(torch.float32), None, ops.ceil32(C))
                y = self.decoder(_conv(self.post_quant_conv, zh))
                outs.append(y[..., :self.decoder.conv_out.out_channels].permute(0, 3, 1, 2).contiguous())
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior
