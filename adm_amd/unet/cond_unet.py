"""MI355X-native conditional super-resolution denoiser behind the reference's module API (SURVEY.md section 8(f) rank 4,
BASELINE configs[4]).

Drop-in for /root/reference/unet/cond_unet_sd.py (`Unet`, one decoder) and /root/reference/unet/cond_unet.py (`Unet`, two
decoders: the class the DIV2K YAML names, configs/super-resolution/div2k_cond_ddm_const_ldm.yaml:42): same constructor
keywords, same ``forward(x, time, mask, ...) -> (C_pred, noise_pred)``, same module tree and therefore the same
state_dict names / shapes / default initialisation (torch.nn modules are used as PARAMETER HOLDERS: nn.Conv2d, nn.Linear,
nn.GroupNorm, nn.BatchNorm2d register exactly the reference's tensors), so reference checkpoints load.  Every forward is
rewritten on the NHWC HIP operators of ``adm_amd.ops`` / ``adm_amd.ops_cond`` -- there is no PyTorch fallback.

Reference lines restated: BasicAttetnionLayer :152-238, RelationNet :240-279, WeightStandardizedConv2d :344-357, LayerNorm
:359-368, Block / ResnetBlock :426-468, LinearAttention :502-530, Attention :532-554, Unet :591-883 (cond_unet.py:823-918 for
the second decoder).

The condition ENCODER ``init_conv_mask`` (torchvision Swin-B / EfficientNet / ResNet with fetched ImageNet weights,
cond_unet_sd.py:637-650) is NOT part of this build: torchvision is absent offline and the weights cannot be fetched.  Its
output -- four feature maps of f, 2f, 4f, 8f channels (f = 128 for Swin-B) at 1/4, 1/8, 1/16, 1/32 of the condition image --
is what ``forward`` takes as ``mask`` (a list of four NCHW tensors), or what a user-supplied ``cond_encoder`` callable
returns for the condition image.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .. import ops_cond as oc

F_COND = {"swin": 128, "resnet": 256, "effnet": 48}


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _pad_vec(v, n):
    if v.shape[0] == n:
        return v
    out = torch.zeros(n, device=v.device, dtype=v.dtype)
    out[: v.shape[0]] = v
    return out


class WeightStandardizedConv2d(nn.Conv2d):
    """3x3 conv on standardised weights; stride-1 only here (the stride-2 variant lives in the unused ConditionEncoder)."""

    def forward(self, x):
        return ops.conv2d(x, oc.weight_standardize(self.weight), self.bias)


class LayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))

    def forward(self, x):
        return oc.layer_norm_c(x, self.g)


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.proj = WeightStandardizedConv2d(dim, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(groups, dim_out)
        self.act = nn.SiLU()
        self.groups = groups

    def forward(self, x, scale_shift=None):
        return ops.group_norm_act(self.proj(x), self.norm.weight, self.norm.bias, scale_shift, silu=True, groups=self.groups,
                                  eps=self.norm.eps)


class ResnetBlock(nn.Module):
    def __init__(self, dim, dim_out, *, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, dim_out * 2)) if time_emb_dim is not None else None
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()

    def forward(self, x, time_emb_silu=None):
        """`time_emb_silu` = SiLU(t_emb), computed once per forward (the reference applies it inside every block's mlp)."""
        ss = None
        if self.mlp is not None and time_emb_silu is not None:
            ss = ops.linear(time_emb_silu, self.mlp[1].weight, self.mlp[1].bias)      # [B, 2 C]: (scale | shift)
        h = self.block1(x, ss)
        h = self.block2(h)
        res = x if isinstance(self.res_conv, nn.Identity) else ops.conv2d(x, self.res_conv.weight, self.res_conv.bias)
        return oc.add(h, res)


class LinearAttention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        if heads != 4 or dim_head != 32:
            raise NotImplementedError("LinearAttention kernels are specialised to 4 heads x 32 (every reference config)")
        self.heads = heads
        hidden = heads * dim_head
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(hidden, dim, 1), LayerNorm(dim))

    def forward(self, x):
        B, H, W, _ = x.shape
        qkv = ops.conv2d(x, self.to_qkv.weight, None)
        out = oc.linear_attention(qkv.reshape(B, H * W, 384)).reshape(B, H, W, 128)
        return self.to_out[1](ops.conv2d(out, self.to_out[0].weight, self.to_out[0].bias))


class Attention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden = heads * dim_head
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden, dim, 1)

    def forward(self, x, residual=None):
        B, H, W, _ = x.shape
        qkv = ops.conv2d(x, self.to_qkv.weight, None)
        hidden = qkv.shape[-1] // 3
        o = oc.self_attention_packed(qkv.reshape(B, H * W, 3 * hidden), self.heads, self.scale).reshape(B, H, W, hidden)
        return ops.conv2d(o, self.to_out.weight, self.to_out.bias, residual)


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)

    def forward(self, x, **kw):
        return self.fn(self.norm(x), **kw)


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x):
        if isinstance(self.fn.fn, Attention):        # the residual add rides in the output conv's epilogue
            return self.fn(x, residual=x)
        return oc.add(self.fn(x), x)


class SpatialAtt(nn.Module):
    """Parameter holder; arithmetic in ops_cond.spatial_att_gate."""

    def __init__(self, in_dim):
        super().__init__()
        self.map = nn.Conv2d(in_dim, 1, 1)
        self.q_conv = nn.Conv2d(1, 1, 1)
        self.k_conv = nn.Conv2d(1, 1, 1)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, drop=0.0):
        super().__init__()
        self.fc1 = nn.Conv2d(in_features, hidden_features or in_features, kernel_size=1)
        self.fc2 = nn.Conv2d(hidden_features or in_features, out_features or in_features, kernel_size=1)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        p = self.drop.p if self.training else 0.0
        h = oc.relu_dropout(ops.conv2d(x, self.fc1.weight, self.fc1.bias), p)
        return oc.dropout(ops.conv2d(h, self.fc2.weight, self.fc2.bias), p)


_POS_CACHE: dict = {}


def pos_embedding_sine(h, w, d, device):
    """PositionEmbeddingSine(normalize=False) for an [*, h, w, d] token grid (cond_unet_sd.py:35-65): a constant of the
    shape, computed once with torch and cached (it carries no parameters and no gradient)."""
    key = (h, w, d, str(device))
    pe = _POS_CACHE.get(key)
    if pe is None:
        npf = d // 2
        y = torch.arange(1, h + 1, dtype=torch.float32, device=device)[:, None].expand(h, w)
        x = torch.arange(1, w + 1, dtype=torch.float32, device=device)[None, :].expand(h, w)
        dim_t = torch.arange(npf, dtype=torch.float32, device=device)
        dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
        px, py = x[..., None] / dim_t, y[..., None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=3).flatten(2)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=3).flatten(2)
        pe = torch.cat((py, px), dim=2).contiguous()
        _POS_CACHE[key] = pe
    return pe


class BasicAttetnionLayer(nn.Module):          # (sic: the reference's spelling, kept for name parity)
    def __init__(self, embed_dim=128, nhead=8, ffn_dim=512, window_size1=(4, 4), window_size2=(1, 1), dropout=0.1):
        super().__init__()
        if window_size1[0] != window_size1[1] or window_size2[0] != window_size2[1]:
            raise NotImplementedError("square pooling windows only (every reference config)")
        self.window_size1, self.window_size2, self.nhead = list(window_size1), list(window_size2), nhead
        self.q_lin = nn.Linear(embed_dim, embed_dim)
        self.k_lin = nn.Linear(embed_dim, embed_dim)
        self.v_lin = nn.Linear(embed_dim, embed_dim)
        self.mlp = Mlp(in_features=embed_dim, hidden_features=ffn_dim, drop=dropout)
        self.concat_conv = nn.Conv2d(2 * embed_dim, embed_dim, 1)
        self.gn = nn.GroupNorm(8, embed_dim)
        self.out_conv = nn.Conv2d(embed_dim, embed_dim, 1)
        for m in self.modules():               # init_weights (cond_unet_sd.py:175-189)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                nn.init.constant_(m.bias, 0.0)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0.0)

    def forward(self, x1, x2):
        """x1: condition feature (queries) [B, H1, W1, C]; x2: UNet feature (keys / values) [B, H2, W2, C]."""
        B, H1, W1, C = x1.shape
        _, H2, W2, _ = x2.shape
        up = oc.bilinear(x1, H2, W2, True)
        sc = ops.conv2d(ops.concat_channels(up, x2), self.concat_conv.weight, self.concat_conv.bias, x2)
        shortcut = ops.group_norm_act(sc, self.gn.weight, self.gn.bias, None, silu=False, groups=8, eps=self.gn.eps)
        x1_s = oc.avg_pool(x1, self.window_size1[0])
        kp = oc.avg_pool(x2, self.window_size2[0])
        hq, wq, hk, wk = x1_s.shape[1], x1_s.shape[2], kp.shape[1], kp.shape[2]
        qg = oc.add(x1_s, pos_embedding_sine(hq, wq, C, x1.device).expand(B, hq, wq, C).contiguous()).reshape(B * hq * wq, C)
        kg = oc.add(kp, pos_embedding_sine(hk, wk, C, x1.device).expand(B, hk, wk, C).contiguous()).reshape(B * hk * wk, C)
        q = ops.linear(qg, self.q_lin.weight, self.q_lin.bias).reshape(B, hq * wq, C)
        k = ops.linear(kg, self.k_lin.weight, self.k_lin.bias).reshape(B, hk * wk, C)
        v = ops.linear(kg, self.v_lin.weight, self.v_lin.bias).reshape(B, hk * wk, C)
        o = oc.mha(q, k, v, self.nhead, 1.0).reshape(B, hq, wq, C)        # no 1/sqrt(d): cond_unet_sd.py:228
        x1_s = oc.add(x1_s, o)
        x1_s = oc.add(x1_s, self.mlp(x1_s))
        x1_s = oc.bilinear(x1_s, H2, W2, True)
        return ops.conv2d(x1_s, self.out_conv.weight, self.out_conv.bias, shortcut)


class RelationNet(nn.Module):
    def __init__(self, in_channel1=128, in_channel2=128, nhead=8, layers=3, embed_dim=128, ffn_dim=512, window_size1=(4, 4),
                 window_size2=(1, 1)):
        super().__init__()
        self.layers = layers
        self.input_conv1 = nn.Sequential(nn.Conv2d(in_channel1, embed_dim, 1), nn.BatchNorm2d(embed_dim, momentum=0.03, eps=0.001))
        self.input_conv2 = nn.Sequential(nn.Conv2d(in_channel2, embed_dim, 1), nn.BatchNorm2d(embed_dim, momentum=0.03, eps=0.001))
        self.attentions = nn.ModuleList([BasicAttetnionLayer(embed_dim=embed_dim, nhead=nhead, ffn_dim=ffn_dim,
                                                             window_size1=window_size1, window_size2=window_size2, dropout=0.1)
                                         for _ in range(layers)])

    def forward(self, cond, feat):
        cond = oc.batch_norm(ops.conv2d(cond, self.input_conv1[0].weight, self.input_conv1[0].bias), self.input_conv1[1], self.training)
        feat = oc.batch_norm(ops.conv2d(feat, self.input_conv2[0].weight, self.input_conv2[0].bias), self.input_conv2[1], self.training)
        for att in self.attentions:
            feat = att(cond, feat)
        return feat


class GaussianFourierProjection(nn.Module):
    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)

    def forward(self, x):
        return oc.fourier_features(x, self.W)


class Downsample(nn.Conv2d):
    """Conv2d(dim, dim_out, 4, 2, 1) (cond_unet_sd.py:341-342)."""

    def __init__(self, dim, dim_out=None):
        super().__init__(dim, dim_out if dim_out is not None else dim, 4, 2, 1)

    def forward(self, x):
        return oc.conv2d_generic(x, self.weight, self.bias, stride=2, pad=1)


def Upsample(dim, dim_out=None):
    """nn.Sequential(nn.Upsample(x2 nearest), Conv2d 3x3): index 1 holds the conv, as in the reference (:335-339); the
    nearest x2 is fused into the conv's loader (ops.conv2d(up=True))."""
    return nn.Sequential(nn.Identity(), nn.Conv2d(dim, dim_out if dim_out is not None else dim, 3, padding=1))


class Unet(nn.Module):
    TWO_DECODERS = True         # this module = reference unet/cond_unet.py (two decoders); cond_unet_sd.Unet pins False

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), cond_in_dim=1, cond_dim=64,
                 cond_dim_mults=(2, 4, 8), channels=1, out_mul=1, self_condition=False, resnet_block_groups=8,
                 learned_variance=False, learned_sinusoidal_cond=False, random_fourier_features=False, learned_sinusoidal_dim=16,
                 window_sizes1=((16, 16), (8, 8), (4, 4), (2, 2)), window_sizes2=((16, 16), (8, 8), (4, 4), (2, 2)),
                 fourier_scale=16, precondition=True, ckpt_path=None, ignore_keys=(), cfg=None, cond_encoder=None, **kwargs):
        super().__init__()
        if self_condition or learned_variance or learned_sinusoidal_cond or random_fourier_features or out_mul != 1:
            raise NotImplementedError("self_condition / learned_variance / learned sinusoidal embeddings / out_mul != 1 are not "
                                      "used by the DDM super-resolution recipe")
        if len(dim_mults) != 4:
            raise NotImplementedError("the condition pyramid has four levels (cond_unet_sd.py:676-679): len(dim_mults) must be 4")
        cond_net = _cfg_get(cfg, "cond_net", kwargs.get("cond_net", "swin"))
        if cond_net not in ("swin", "resnet"):
            raise NotImplementedError(f"cond_net {cond_net!r}: only the f, 2f, 4f, 8f pyramids (swin, resnet) are implemented")
        if _cfg_get(cfg, "cond_pe", kwargs.get("cond_pe", False)):
            raise NotImplementedError("cond_pe is False in every reference config")
        f = F_COND[cond_net]
        self.f_cond = f
        self.channels, self.self_condition, self.precondition = channels, self_condition, precondition
        self.two_decoders = self.TWO_DECODERS
        # the condition encoder (torchvision backbone + fetched weights) is supplied by the caller, see the module docstring
        self.init_conv_mask = cond_encoder
        init_dim = init_dim if init_dim is not None else dim
        if dim % 32 or init_dim % 32:
            raise NotImplementedError("channel widths must be multiples of 32")
        self.init_conv = nn.Sequential(nn.Conv2d(channels + f, init_dim, 7, padding=3),
                                       nn.GroupNorm(num_groups=min(init_dim // 4, 8), num_channels=init_dim))
        dims = [init_dim, *[dim * m for m in dim_mults]]
        rev = dims[::-1]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.projects = nn.ModuleList([nn.Conv2d(f * 2 ** i, dims[i], 1) for i in range(4)])
        time_dim = dim * 4
        self.time_mlp = nn.Sequential(GaussianFourierProjection(dim // 2, scale=fourier_scale), nn.Linear(dim, time_dim), nn.GELU(),
                                      nn.Linear(time_dim, time_dim))
        block = lambda a, b: ResnetBlock(a, b, time_emb_dim=time_dim, groups=resnet_block_groups)
        n = len(in_out)
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        self.relation_layers_down = nn.ModuleList([])
        self.relation_layers_up = nn.ModuleList([])
        if self.two_decoders:           # registration order of cond_unet.py:710-714 (= the order of a reference optimiser state)
            self.ups2 = nn.ModuleList([])
        self.relation_layers_up2 = nn.ModuleList([])
        w1, w2 = [list(w) for w in window_sizes1], [list(w) for w in window_sizes2]
        for ind, (di, do) in enumerate(in_out):
            last = ind >= n - 1
            self.downs.append(nn.ModuleList([block(di, di), block(di, di), Residual(PreNorm(di, LinearAttention(di))),
                                             Downsample(di, do) if not last else nn.Conv2d(di, do, 3, padding=1)]))
            self.relation_layers_down.append(RelationNet(dims[ind], dims[ind], nhead=8, layers=1, embed_dim=dims[ind],
                                                         ffn_dim=dims[ind] * 2, window_size1=w1[ind], window_size2=w2[ind]))
        mid = dims[-1]
        self.mid_block1 = block(mid, mid)
        self.mid_attn = Residual(PreNorm(mid, Attention(mid)))
        self.mid_block2 = block(mid, mid)

        def decouple():
            return nn.Sequential(nn.GroupNorm(num_groups=min(mid // 4, 8), num_channels=mid), nn.Conv2d(mid, mid, 3, padding=1),
                                 SpatialAtt(mid))

        self.decouple1 = decouple()
        if self.two_decoders:
            self.decouple2 = decouple()
        for ind, (di, do) in enumerate(reversed(in_out)):
            last = ind == n - 1
            mk = lambda: nn.ModuleList([block(do + di, do), block(do + di, do), Residual(PreNorm(do, LinearAttention(do))),
                                        Upsample(do, di) if not last else nn.Conv2d(do, di, 3, padding=1)])
            rel = lambda: RelationNet(rev[ind + 1], rev[ind], nhead=8, layers=1, embed_dim=rev[ind], ffn_dim=rev[ind] * 2,
                                      window_size1=w1[::-1][ind], window_size2=w2[::-1][ind])
            self.ups.append(mk())
            self.relation_layers_up.append(rel())
            if self.two_decoders:
                self.ups2.append(mk())
                self.relation_layers_up2.append(rel())
        self.out_dim = out_dim if out_dim is not None else channels
        self.final_res_block = block(dim * 2, dim)
        self.final_conv = nn.Conv2d(dim, self.out_dim * out_mul, 1)
        if self.two_decoders:
            self.final_res_block2 = block(dim * 2, dim)
            self.final_conv2 = nn.Conv2d(dim, self.out_dim, 1)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=()):
        sd = torch.load(path, map_location="cpu", weights_only=True)["model"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys) or k.startswith("init_conv_mask."):
                del sd[k]
        msg = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}: {msg}")

    # -------------------------------------------------------------------------------------------- forward
    def cond_features(self, mask):
        if isinstance(mask, (list, tuple)):
            hm = list(mask)
        elif self.init_conv_mask is not None:
            hm = list(self.init_conv_mask(mask))
        else:
            raise RuntimeError("cond_unet.Unet: no condition encoder.  Pass the four encoder feature maps as `mask` (list of "
                               "NCHW tensors with f, 2f, 4f, 8f channels) or construct the model with cond_encoder=<callable>; "
                               "the reference's torchvision Swin-B and its ImageNet weights are not available offline")
        if len(hm) != 4 or any(h.shape[1] != self.f_cond * 2 ** i for i, h in enumerate(hm)):
            raise RuntimeError(f"condition features must be 4 maps with {[self.f_cond * 2 ** i for i in range(4)]} channels")
        return [ops.nchw_to_nhwc(h.to(torch.float32), None, h.shape[1]) for h in hm]

    def _decouple(self, seq, x):
        gn, conv, sa = seq[0], seq[1], seq[2]
        h = ops.group_norm_act(x, gn.weight, gn.bias, None, silu=False, groups=gn.num_groups, eps=gn.eps)
        h = ops.conv2d(h, conv.weight, conv.bias)
        att = ops.conv2d(h, sa.map.weight, sa.map.bias)
        qk = torch.cat([sa.q_conv.weight.reshape(1), sa.q_conv.bias, sa.k_conv.weight.reshape(1), sa.k_conv.bias])
        return oc.spatial_att_gate(att, qk, h, x)                # softsign(gate) * h + x

    def _decode(self, x, ups, relations, h, hm, r, te, final_block, final_conv):
        hs, hms = list(h), list(hm)
        n = len(ups)
        for i, ((b1, b2, attn, up), rel) in enumerate(zip(ups, relations)):
            x = b1(ops.concat_channels(x, hs.pop()), te)
            x = rel(hms.pop(), x)
            x = b2(ops.concat_channels(x, hs.pop()), te)
            x = attn(x)
            if i < n - 1:
                x = ops.conv2d(x, up[1].weight, up[1].bias, up=True)
            else:
                x = ops.conv2d(x, up.weight, up.bias)
        x = final_block(ops.concat_channels(x, r), te)
        return ops.conv2d(x, final_conv.weight, final_conv.bias)

    def forward(self, x, time, mask, x_self_cond=None, sigma_max=1, *args, **kwargs):
        dev = x.device
        time = torch.as_tensor(time, device=dev).to(torch.float32).reshape(-1)
        B, Cx, H, W = x.shape
        if time.numel() == 1 and B > 1:
            time = time.expand(B).contiguous()
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        x = x.contiguous()
        hm = self.cond_features(mask)
        # stem input: [latent (channels) | bilinear(hm[0]) (f)] in ONE NHWC tensor padded to a multiple of 32 channels
        cin = ops.ceil32(Cx + self.f_cond)
        xin = ops.nchw_to_nhwc(x, None, cin)
        oc.bilinear_into(hm[0], xin, Cx, False)
        h0 = oc.conv2d_generic(xin, self.init_conv[0].weight, self.init_conv[0].bias, stride=1, pad=3)
        gn = self.init_conv[1]
        xx = ops.group_norm_act(h0, gn.weight, gn.bias, None, silu=False, groups=gn.num_groups, eps=gn.eps)
        r = xx
        emb = self.time_mlp[0](time.log())
        emb = oc.gelu(ops.linear(emb, self.time_mlp[1].weight, self.time_mlp[1].bias))
        te = ops.silu(ops.linear(emb, self.time_mlp[3].weight, self.time_mlp[3].bias))     # SiLU(t_emb), shared by all blocks
        hm = [ops.conv2d(hm[i], p.weight, p.bias) for i, p in enumerate(self.projects)]
        h = []
        for (b1, b2, attn, down), rel, hmi in zip(self.downs, self.relation_layers_down, hm):
            xx = b1(xx, te)
            h.append(xx)
            xx = rel(hmi, xx)
            xx = b2(xx, te)
            xx = attn(xx)
            h.append(xx)
            xx = down(xx) if isinstance(down, Downsample) else ops.conv2d(xx, down.weight, down.bias)
        xx = self.mid_block1(xx, te)
        xx = self.mid_attn(xx)
        xm = self.mid_block2(xx, te)
        f1 = self._decode(self._decouple(self.decouple1, xm), self.ups, self.relation_layers_up, h, hm, r, te,
                          self.final_res_block, self.final_conv)
        t = time
        one, zero = torch.ones_like(t), torch.zeros_like(t)
        if self.precondition:        # x1 = (t - 1) x + t / sqrt(t + 1) F1      (cond_unet_sd.py:810-813, 879)
            x1 = ops.precond_out(f1, x, (t - 1).contiguous(), (t / (t + 1).sqrt()).contiguous())
        else:
            x1 = ops.precond_out(f1, x, zero, one)
        if self.two_decoders:        # cond_unet.py:902-916
            f2 = self._decode(self._decouple(self.decouple2, xm), self.ups2, self.relation_layers_up2, h, hm, r, te,
                              self.final_res_block2, self.final_conv2)
            if self.precondition:
                x2 = ops.precond_out(f2, x, t.sqrt().contiguous(), ((1 - t).sqrt() / (1 + t).sqrt()).contiguous())
            else:
                x2 = ops.precond_out(f2, x, zero, one)
        else:                        # x2 = (x - (t - 1) x1) / sqrt(t)           (cond_unet_sd.py:880-882)
            x2 = ops.axpby_batch(x1, x, (1 / t.sqrt()).contiguous(), (-(t - 1) / t.sqrt()).contiguous())
        return x1, x2
