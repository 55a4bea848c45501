"""Oracle (CPU, plain PyTorch) restatement of the reference's preconditioned two-output UNet.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Written as pure functions over a flat
``{name: tensor}`` state dict that uses the reference's own key names, so a reference
checkpoint / state_dict can be fed in directly.

Reference being restated (file:line are into /root/reference):
  * layer primitives        unet/uncond_unet.py:53-129   (Linear, Conv2d incl. up/down, GroupNorm)
  * UNetBlock.forward       unet/uncond_unet.py:189-211
  * PositionalEmbedding     unet/uncond_unet.py:217-230
  * SpatialAtt              unet/uncond_unet.py:19-37
  * DhariwalUNet            unet/uncond_unet.py:450-581  (two decoders)
                            unet/uncond_unet_sd.py:450-551 (single decoder)
                            unet/uncond_unet_sd_3.py:547-555 (skip-tuning ratios)
  * EDMPrecond.forward      unet/uncond_unet.py:614-635, uncond_unet_2.py:623-627,
                            uncond_unet_sd.py:591-605, uncond_unet_sd_2.py:592-606,
                            uncond_unet_sd_3.py:598-612
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

VARIANTS = ("uncond_unet", "uncond_unet_2", "uncond_unet_sd", "uncond_unet_sd_2", "uncond_unet_sd_3")


def default_cfg(**over) -> dict:
    """UNet hyper-parameters of configs/cifar10/ddm_uncond_const_uncond_unet.yaml:18-31."""
    cfg = dict(img_resolution=32, img_channels=3, model_channels=192, channel_mult=[1, 2, 2, 2],
               channel_mult_emb=4, num_blocks=3, attn_resolutions=[16, 8], dropout=0.1,
               augment_dim=9, variant="uncond_unet")
    cfg.update(over)
    assert cfg["variant"] in VARIANTS
    return cfg


def two_decoders(variant: str) -> bool:
    return variant in ("uncond_unet", "uncond_unet_2")


# ----------------------------------------------------------------------------------------------
# Architecture enumeration (unet/uncond_unet.py:482-542)
# ----------------------------------------------------------------------------------------------

def block_specs(cfg: dict) -> Dict[str, List[dict]]:
    """Returns {'enc': [...], 'dec': [...]} of block descriptors in execution order.

    Each descriptor: name, kind ('conv' for the stem, else 'block'), cin, cout, res (output
    resolution), up, down, attn.  dec2 (when present) has the same list as dec."""
    mc, mults, nb = cfg["model_channels"], cfg["channel_mult"], cfg["num_blocks"]
    R, attn_res = cfg["img_resolution"], cfg["attn_resolutions"]
    enc: List[dict] = []
    cout = cfg["img_channels"]
    for level, mult in enumerate(mults):
        res = R >> level
        if level == 0:
            cin, cout = cout, mc * mult
            enc.append(dict(name=f"{res}x{res}_conv", kind="conv", cin=cin, cout=cout, res=res,
                            up=False, down=False, attn=False))
        else:
            enc.append(dict(name=f"{res}x{res}_down", kind="block", cin=cout, cout=cout, res=res,
                            up=False, down=True, attn=False))
        for idx in range(nb):
            cin, cout = cout, mc * mult
            enc.append(dict(name=f"{res}x{res}_block{idx}", kind="block", cin=cin, cout=cout, res=res,
                            up=False, down=False, attn=res in attn_res))
    skips = [b["cout"] for b in enc]
    dec: List[dict] = []
    for level, mult in reversed(list(enumerate(mults))):
        res = R >> level
        if level == len(mults) - 1:
            dec.append(dict(name=f"{res}x{res}_in0", kind="block", cin=cout, cout=cout, res=res,
                            up=False, down=False, attn=True))
            dec.append(dict(name=f"{res}x{res}_in1", kind="block", cin=cout, cout=cout, res=res,
                            up=False, down=False, attn=False))
        else:
            dec.append(dict(name=f"{res}x{res}_up", kind="block", cin=cout, cout=cout, res=res,
                            up=True, down=False, attn=False))
        for idx in range(nb + 1):
            cin = cout + skips.pop()
            cout = mc * mult
            dec.append(dict(name=f"{res}x{res}_block{idx}", kind="block", cin=cin, cout=cout, res=res,
                            up=False, down=False, attn=res in attn_res))
    return dict(enc=enc, dec=dec, bottleneck=enc[-1]["cout"], head=cout)


def param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """Every state_dict entry (parameters AND buffers) of EDMPrecond at ``cfg``, with the
    reference's names ('model.' prefix = EDMPrecond.model) in the reference's registration order."""
    mc = cfg["model_channels"]
    emb = mc * cfg["channel_mult_emb"]
    specs = block_specs(cfg)
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(prefix, cin, cout, k, resample=False):
        if k:
            out[prefix + ".weight"] = (cout, cin, k, k)
            out[prefix + ".bias"] = (cout,)
        if resample:
            out[prefix + ".resample_filter"] = (1, 1, 2, 2)

    def norm(prefix, c):
        out[prefix + ".weight"] = (c,)
        out[prefix + ".bias"] = (c,)

    def lin(prefix, cin, cout, bias=True):
        out[prefix + ".weight"] = (cout, cin)
        if bias:
            out[prefix + ".bias"] = (cout,)

    def block(prefix, b):
        rs = b["up"] or b["down"]
        norm(prefix + ".norm0", b["cin"])
        conv(prefix + ".conv0", b["cin"], b["cout"], 3, rs)
        lin(prefix + ".affine", emb, 2 * b["cout"])
        norm(prefix + ".norm1", b["cout"])
        conv(prefix + ".conv1", b["cout"], b["cout"], 3)
        if b["cin"] != b["cout"] or rs:
            conv(prefix + ".skip", b["cin"], b["cout"], 1 if b["cin"] != b["cout"] else 0, rs)
        if b["attn"]:
            norm(prefix + ".norm2", b["cout"])
            conv(prefix + ".qkv", b["cout"], 3 * b["cout"], 1)
            conv(prefix + ".proj", b["cout"], b["cout"], 1)

    if cfg["augment_dim"]:
        lin("model.map_augment", cfg["augment_dim"], mc, bias=False)
    lin("model.map_layer0", mc, emb)
    lin("model.map_layer1", emb, emb)
    for b in specs["enc"]:
        if b["kind"] == "conv":
            conv("model.enc." + b["name"], b["cin"], b["cout"], 3)
        else:
            block("model.enc." + b["name"], b)
    cb = specs["bottleneck"]
    decouples = ["decouple1", "decouple2"] if two_decoders(cfg["variant"]) else ["decouple1"]
    for d in decouples:
        out[f"model.{d}.0.weight"] = (cb, cb, 3, 3)
        out[f"model.{d}.0.bias"] = (cb,)
        out[f"model.{d}.1.map.weight"] = (1, cb, 1, 1)
        out[f"model.{d}.1.map.bias"] = (1,)
        for n in ("q_conv", "k_conv"):
            out[f"model.{d}.1.{n}.weight"] = (1, 1, 1, 1)
            out[f"model.{d}.1.{n}.bias"] = (1,)
    for b in specs["dec"]:
        block("model.dec." + b["name"], b)
    norm("model.out_norm", specs["head"])
    conv("model.out_conv", specs["head"], cfg["img_channels"], 3)
    if two_decoders(cfg["variant"]):
        for b in specs["dec"]:
            block("model.dec2." + b["name"], b)
        norm("model.out_norm2", specs["head"])
        conv("model.out_conv2", specs["head"], cfg["img_channels"], 3)
    return out


# ----------------------------------------------------------------------------------------------
# Primitives
# ----------------------------------------------------------------------------------------------

def _conv(sd, prefix: str, x: Tensor, up=False, down=False) -> Tensor:
    """unet/uncond_unet.py:91-113 with resample_filter=[1,1], fused_resample=False."""
    c = x.shape[1]
    if up:    # depthwise conv_transpose2d with an all-ones 2x2 (f*4), stride 2 == nearest x2
        x = F.conv_transpose2d(x, torch.ones(c, 1, 2, 2, dtype=x.dtype), groups=c, stride=2)
    if down:  # depthwise 2x2 box /4, stride 2
        x = F.conv2d(x, torch.full((c, 1, 2, 2), 0.25, dtype=x.dtype), groups=c, stride=2)
    w = sd.get(prefix + ".weight")
    if w is not None:
        x = F.conv2d(x, w.to(x.dtype), padding=w.shape[-1] // 2)
    b = sd.get(prefix + ".bias")
    if b is not None:
        x = x + b.to(x.dtype).reshape(1, -1, 1, 1)
    return x


def _gn(sd, prefix: str, x: Tensor, eps=1e-5) -> Tensor:
    """unet/uncond_unet.py:119-129."""
    c = x.shape[1]
    return F.group_norm(x, min(32, c // 4), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _linear(sd, prefix: str, x: Tensor) -> Tensor:
    y = x @ sd[prefix + ".weight"].t()
    b = sd.get(prefix + ".bias")
    return y if b is None else y + b


def positional_embedding(x: Tensor, num_channels: int, max_positions=10000) -> Tensor:
    """unet/uncond_unet.py:224-230 (endpoint=False)."""
    half = num_channels // 2
    freqs = torch.arange(half, dtype=torch.float32) / half
    freqs = (1.0 / max_positions) ** freqs
    ang = torch.outer(x, freqs.to(x.dtype))
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def attention_core(qkv: Tensor, heads: int) -> Tensor:
    """unet/uncond_unet.py:205-208.  qkv [B,3C,H,W]; channel order (head, c, {q,k,v})."""
    B, C3, H, W = qkv.shape
    C = C3 // 3
    q, k, v = qkv.reshape(B * heads, C // heads, 3, H * W).unbind(2)
    w = torch.einsum("ncq,nck->nqk", q, k / math.sqrt(k.shape[1])).softmax(dim=2)
    a = torch.einsum("nqk,nck->ncq", w, v)
    return a.reshape(B, C, H, W)


def unet_block(sd, prefix: str, b: dict, x: Tensor, emb: Tensor, dropout=0.0, training=False) -> Tensor:
    """unet/uncond_unet.py:189-211 (adaptive_scale=True, skip_scale=1)."""
    orig = x
    x = _conv(sd, prefix + ".conv0", F.silu(_gn(sd, prefix + ".norm0", x)), up=b["up"], down=b["down"])
    params = _linear(sd, prefix + ".affine", emb)[:, :, None, None]
    scale, shift = params.chunk(2, dim=1)
    x = F.silu(torch.addcmul(shift, _gn(sd, prefix + ".norm1", x), scale + 1))
    x = _conv(sd, prefix + ".conv1", F.dropout(x, p=dropout, training=training))
    has_skip = (b["cin"] != b["cout"]) or b["up"] or b["down"]
    x = x + (_conv(sd, prefix + ".skip", orig, up=b["up"], down=b["down"]) if has_skip else orig)
    if b["attn"]:
        heads = b["cout"] // 64
        a = attention_core(_conv(sd, prefix + ".qkv", _gn(sd, prefix + ".norm2", x)), heads)
        x = _conv(sd, prefix + ".proj", a) + x
    return x


def spatial_att(sd, prefix: str, x: Tensor) -> Tensor:
    """unet/uncond_unet.py:27-37."""
    B, _, H, W = x.shape
    att = _conv(sd, prefix + ".map", x)                       # [B,1,H,W]
    q = _conv(sd, prefix + ".q_conv", att).reshape(B, H * W, 1)
    k = _conv(sd, prefix + ".k_conv", att).reshape(B, 1, H * W)
    a = att.reshape(B, H * W, 1)
    a = torch.softmax(q @ k, dim=-1) @ a                      # [B,HW,1]
    return F.softsign(a.reshape(B, 1, H, W)) * x


def decouple(sd, prefix: str, x: Tensor) -> Tensor:
    """decouple{1,2}(x) + x : unet/uncond_unet.py:500-507, 566-567."""
    return spatial_att(sd, prefix + ".1", _conv(sd, prefix + ".0", x)) + x


def time_embedding(sd, cfg: dict, noise_labels: Tensor, augment_labels: Optional[Tensor]) -> Tensor:
    """unet/uncond_unet.py:546-556 (label_dim = 0)."""
    emb = positional_embedding(noise_labels, cfg["model_channels"])
    if cfg["augment_dim"] and augment_labels is not None:
        emb = emb + _linear(sd, "model.map_augment", augment_labels)
    emb = F.silu(_linear(sd, "model.map_layer0", emb))
    return F.silu(_linear(sd, "model.map_layer1", emb))


def dhariwal_unet(sd, cfg: dict, x: Tensor, noise_labels: Tensor, augment_labels=None, training=False):
    """unet/uncond_unet.py:544-581.  Returns (F_x, F_y) for two-decoder variants, (F_x, None) else."""
    specs = block_specs(cfg)
    p = cfg["dropout"]
    emb = time_embedding(sd, cfg, noise_labels, augment_labels)
    skips = []
    for b in specs["enc"]:
        pre = "model.enc." + b["name"]
        x = _conv(sd, pre, x) if b["kind"] == "conv" else unet_block(sd, pre, b, x, emb, p, training)
        skips.append(x)

    def run_decoder(dname, x1, out_norm, out_conv, ratios=None):
        stack = list(skips)
        r = list(ratios) if ratios is not None else None
        for b in specs["dec"]:
            if x1.shape[1] != b["cin"]:
                s = stack.pop()
                if r is not None:
                    s = s * r.pop()
                x1 = torch.cat([x1, s], dim=1)
            x1 = unet_block(sd, f"model.{dname}." + b["name"], b, x1, emb, p, training)
        return _conv(sd, out_conv, F.silu(_gn(sd, out_norm, x1)))

    ratios = None
    if cfg["variant"] == "uncond_unet_sd_3":  # uncond_unet_sd_3.py:547-555
        n = len(skips)
        ratios = [0.5 + 0.5 * i / (n - 1) for i in range(n)][::-1]
    f_x = run_decoder("dec", decouple(sd, "model.decouple1", x), "model.out_norm", "model.out_conv", ratios)
    f_y = None
    if two_decoders(cfg["variant"]):
        f_y = run_decoder("dec2", decouple(sd, "model.decouple2", x), "model.out_norm2", "model.out_conv2")
    return f_x, f_y


def precond_coeffs(variant: str, sigma: Tensor):
    """(c_skip1, c_out1, c_skip2, c_out2, c_in, c_noise) for sigma shaped [-1,1,1,1].
    unet/uncond_unet.py:621-626 ('const' family) vs uncond_unet_2.py:623-627 ('const_2' family)."""
    if variant in ("uncond_unet", "uncond_unet_sd"):
        den = sigma ** 2 - sigma + 1
        c_skip1 = (sigma - 1) / den
        c_skip2 = sigma.sqrt() / den
        c_out1 = torch.sqrt(sigma / den)
        c_out2 = (1 - sigma) / den.sqrt()
        c_in = 1 / torch.sqrt((1 - sigma) ** 2 + sigma)
    else:
        den = sigma ** 2 + (sigma - 1) ** 2
        c_skip1 = (sigma - 1) / den
        c_out1 = sigma / den.sqrt()
        c_skip2 = sigma / den
        c_out2 = (1 - sigma) / den.sqrt()
        c_in = 1 / den.sqrt()
    return c_skip1, c_out1, c_skip2, c_out2, c_in, sigma.log()


def edm_precond(sd, cfg: dict, x: Tensor, sigma: Tensor, augment_labels=None, training=False):
    """EDMPrecond.forward (precondition=True).  x any float dtype NCHW, sigma [B] or 0-dim.
    Returns (D_x, D_y) fp32."""
    v = cfg["variant"]
    x = x.to(torch.float32)
    sigma = sigma.to(torch.float32).reshape(-1, 1, 1, 1)
    c_skip1, c_out1, c_skip2, c_out2, c_in, c_noise = precond_coeffs(v, sigma)
    f_x, f_y = dhariwal_unet(sd, cfg, c_in * x, c_noise.flatten(), augment_labels, training)
    d_x = c_skip1 * x + c_out1 * f_x
    if two_decoders(v):
        d_y = c_skip2 * x + c_out2 * f_y
    elif v == "uncond_unet_sd":
        d_y = (x - (sigma - 1) * d_x) / sigma.sqrt()
    else:
        d_y = (x - (sigma - 1) * d_x) / sigma
    return d_x, d_y
