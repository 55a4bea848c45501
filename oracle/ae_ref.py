"""Oracle (CPU, plain PyTorch) restatement of the reference's KL autoencoder ("first stage") and of the latent
variant of the analytic-schedule wrapper.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity PINNED: tools/make_golden_latent.py runs these functions
against the imported reference classes (ddm.encoder_decoder.Encoder / Decoder, ddm.ddm_const_2.LatentDiffusion) on
identical closed-form inputs and commits the reference outputs as tests/golden/g10_autoencoder.npz /
g11_latent.npz.

The network is a pure function of a flat state dict that uses the reference's parameter names
(``encoder.down.0.block.1.norm2.weight`` ...), so a reference checkpoint loads as is.
  Encoder            /root/reference/ddm/encoder_decoder.py:386-479
  Decoder            /root/reference/ddm/encoder_decoder.py:482-587
  ResnetBlock        :99-160     AttnBlock :169-213     Downsample :78-96     Upsample :60-75
  AutoencoderKL      :894-950 (encode = quant_conv(encoder(x)) -> moments; decode = decoder(post_quant_conv(z)))
  DiagonalGaussianDistribution :854-892
  LatentDiffusion    /root/reference/ddm/ddm_const_2.py:393-737
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def ae_cfg(ch=128, ch_mult=(1, 2, 4), num_res_blocks=2, z_channels=3, embed_dim=3, in_channels=3, out_ch=3,
           resolution=(256, 256), attn_resolutions=()):
    """KL-f4 defaults = configs/celebahq/celeb_uncond_ddm_const2_unet_ldm.yaml:23-40."""
    return dict(ch=ch, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks, z_channels=z_channels, embed_dim=embed_dim,
                in_channels=in_channels, out_ch=out_ch, resolution=tuple(resolution),
                attn_resolutions=tuple(tuple(a) for a in attn_resolutions))


def _res_shapes(p: str, cin: int, cout: int) -> Dict[str, Tuple[int, ...]]:
    s = {f"{p}.norm1.weight": (cin,), f"{p}.norm1.bias": (cin,),
         f"{p}.conv1.weight": (cout, cin, 3, 3), f"{p}.conv1.bias": (cout,),
         f"{p}.norm2.weight": (cout,), f"{p}.norm2.bias": (cout,),
         f"{p}.conv2.weight": (cout, cout, 3, 3), f"{p}.conv2.bias": (cout,)}
    if cin != cout:
        s[f"{p}.nin_shortcut.weight"] = (cout, cin, 1, 1)
        s[f"{p}.nin_shortcut.bias"] = (cout,)
    return s


def _attn_shapes(p: str, c: int) -> Dict[str, Tuple[int, ...]]:
    s = {f"{p}.norm.weight": (c,), f"{p}.norm.bias": (c,)}
    for n in ("q", "k", "v", "proj_out"):
        s[f"{p}.{n}.weight"] = (c, c, 1, 1)
        s[f"{p}.{n}.bias"] = (c,)
    return s


def _level_has_attn(cfg, res) -> bool:
    return tuple(res) in cfg["attn_resolutions"]


def param_shapes(cfg) -> Dict[str, Tuple[int, ...]]:
    """Names / shapes / order of AutoencoderKL.state_dict() minus the `loss.*` (LPIPS + discriminator) entries."""
    ch, mult, nrb = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"]
    nres = len(mult)
    in_mult = (1,) + tuple(mult)
    s: Dict[str, Tuple[int, ...]] = {}
    # encoder
    s["encoder.conv_in.weight"] = (ch, cfg["in_channels"], 3, 3); s["encoder.conv_in.bias"] = (ch,)
    res = tuple(cfg["resolution"])
    bi = ch
    for lv in range(nres):
        bi, bo = ch * in_mult[lv], ch * mult[lv]
        for b in range(nrb):
            s.update(_res_shapes(f"encoder.down.{lv}.block.{b}", bi, bo))
            bi = bo
            if _level_has_attn(cfg, res):
                s.update(_attn_shapes(f"encoder.down.{lv}.attn.{b}", bi))
        if lv != nres - 1:
            s[f"encoder.down.{lv}.downsample.conv.weight"] = (bi, bi, 3, 3)
            s[f"encoder.down.{lv}.downsample.conv.bias"] = (bi,)
            res = (res[0] // 2, res[1] // 2)
    s.update(_res_shapes("encoder.mid.block_1", bi, bi))
    s.update(_attn_shapes("encoder.mid.attn_1", bi))
    s.update(_res_shapes("encoder.mid.block_2", bi, bi))
    s["encoder.norm_out.weight"] = (bi,); s["encoder.norm_out.bias"] = (bi,)
    s["encoder.conv_out.weight"] = (2 * cfg["z_channels"], bi, 3, 3); s["encoder.conv_out.bias"] = (2 * cfg["z_channels"],)
    # decoder (registration order: conv_in, mid, up (module list, level 0 first), norm_out, conv_out)
    bi = ch * mult[nres - 1]
    s["decoder.conv_in.weight"] = (bi, cfg["z_channels"], 3, 3); s["decoder.conv_in.bias"] = (bi,)
    s.update(_res_shapes("decoder.mid.block_1", bi, bi))
    s.update(_attn_shapes("decoder.mid.attn_1", bi))
    s.update(_res_shapes("decoder.mid.block_2", bi, bi))
    res = (cfg["resolution"][0] // 2 ** (nres - 1), cfg["resolution"][1] // 2 ** (nres - 1))
    up: Dict[int, Dict[str, Tuple[int, ...]]] = {}
    for lv in reversed(range(nres)):
        d: Dict[str, Tuple[int, ...]] = {}
        bo = ch * mult[lv]
        for b in range(nrb + 1):
            d.update(_res_shapes(f"decoder.up.{lv}.block.{b}", bi, bo))
            bi = bo
            if _level_has_attn(cfg, res):
                d.update(_attn_shapes(f"decoder.up.{lv}.attn.{b}", bi))
        if lv != 0:
            d[f"decoder.up.{lv}.upsample.conv.weight"] = (bi, bi, 3, 3)
            d[f"decoder.up.{lv}.upsample.conv.bias"] = (bi,)
            res = (res[0] * 2, res[1] * 2)
        up[lv] = d
    for lv in range(nres):
        s.update(up[lv])
    s["decoder.norm_out.weight"] = (bi,); s["decoder.norm_out.bias"] = (bi,)
    s["decoder.conv_out.weight"] = (cfg["out_ch"], bi, 3, 3); s["decoder.conv_out.bias"] = (cfg["out_ch"],)
    s["quant_conv.weight"] = (2 * cfg["embed_dim"], 2 * cfg["z_channels"], 1, 1); s["quant_conv.bias"] = (2 * cfg["embed_dim"],)
    s["post_quant_conv.weight"] = (cfg["z_channels"], cfg["embed_dim"], 1, 1); s["post_quant_conv.bias"] = (cfg["z_channels"],)
    return s


# ------------------------------------------------------------------------------------------------
def _norm(sd, p, x):
    """Normalize = GroupNorm(32 groups, eps 1e-6, affine)  (encoder_decoder.py:56-57)."""
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)        # encoder_decoder.py:51-53


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def resnet_block(sd, p, x):
    """encoder_decoder.py:138-160 with temb = None, dropout inactive."""
    h = _conv(sd, p + ".conv1", _swish(_norm(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", _swish(_norm(sd, p + ".norm2", h)))
    if p + ".nin_shortcut.weight" in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def attn_block(sd, p, x):
    """Single-head spatial self-attention, head dim = C (encoder_decoder.py:190-213)."""
    h = _norm(sd, p + ".norm", x)
    q, k, v = (_conv(sd, f"{p}.{n}", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    o = torch.bmm(v.reshape(b, c, hh * ww), w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", o, padding=0)


def downsample(sd, p, x):
    """pad (0,1,0,1) then 3x3 stride-2 conv without padding (encoder_decoder.py:78-96)."""
    return _conv(sd, p + ".conv", F.pad(x, (0, 1, 0, 1), mode="constant", value=0), stride=2, padding=0)


def upsample(sd, p, x):
    """nearest x2 then 3x3 conv (encoder_decoder.py:60-75)."""
    return _conv(sd, p + ".conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))


def encoder(sd, cfg, x):
    nres, nrb = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    h = _conv(sd, "encoder.conv_in", x)
    for lv in range(nres):
        for b in range(nrb):
            h = resnet_block(sd, f"encoder.down.{lv}.block.{b}", h)
            if f"encoder.down.{lv}.attn.{b}.norm.weight" in sd:
                h = attn_block(sd, f"encoder.down.{lv}.attn.{b}", h)
        if lv != nres - 1:
            h = downsample(sd, f"encoder.down.{lv}.downsample", h)
    h = resnet_block(sd, "encoder.mid.block_1", h)
    h = attn_block(sd, "encoder.mid.attn_1", h)
    h = resnet_block(sd, "encoder.mid.block_2", h)
    return _conv(sd, "encoder.conv_out", _swish(_norm(sd, "encoder.norm_out", h)))


def decoder(sd, cfg, z):
    nres, nrb = len(cfg["ch_mult"]), cfg["num_res_blocks"]
    h = _conv(sd, "decoder.conv_in", z)
    h = resnet_block(sd, "decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h)
    for lv in reversed(range(nres)):
        for b in range(nrb + 1):
            h = resnet_block(sd, f"decoder.up.{lv}.block.{b}", h)
            if f"decoder.up.{lv}.attn.{b}.norm.weight" in sd:
                h = attn_block(sd, f"decoder.up.{lv}.attn.{b}", h)
        if lv != 0:
            h = upsample(sd, f"decoder.up.{lv}.upsample", h)
    return _conv(sd, "decoder.conv_out", _swish(_norm(sd, "decoder.norm_out", h)))


def encode_moments(sd, cfg, x):
    """AutoencoderKL.encode up to the posterior's parameters (encoder_decoder.py:937-941)."""
    return _conv(sd, "quant_conv", encoder(sd, cfg, x), padding=0)


def posterior_sample(moments, eps: Optional[Tensor]):
    """DiagonalGaussianDistribution(moments).sample() with the N(0,1) draw injected; eps None -> mode()
    (encoder_decoder.py:855-867, 891-892)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    if eps is None:
        return mean
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * eps


def decode(sd, cfg, z):
    """AutoencoderKL.decode (encoder_decoder.py:943-946)."""
    return decoder(sd, cfg, _conv(sd, "post_quant_conv", z, padding=0))


# ------------------------------------------------------------------------------------------------
# LatentDiffusion (const_2 schedule): /root/reference/ddm/ddm_const_2.py:393-737
# ------------------------------------------------------------------------------------------------
def _bc(t, ref):
    return t.reshape(ref.shape[0], *((1,) * (ref.dim() - 1)))


def std_scale_factor(z: Tensor) -> Tensor:
    """on_train_batch_start: scale_factor = 1 / std of the first batch's encodings (ddm_const_2.py:478-489)."""
    return 1.0 / z.flatten().std()


def latent_p_losses(model_fn: Callable, z0: Tensor, t: Tensor, noise: Tensor, eps: float, weighting_loss: bool = True):
    """ddm_const_2.py:527-596 with use_l1 = False, use_disloss = False: the weighted SSE of the pixel-space wrapper
    plus an L1 reconstruction term  loss_vlb = sum|x_rec - z0| * rec_weight  (no LPIPS in latent space)."""
    C = -1 * z0
    time = _bc(t, C)
    x_noisy = z0 + C * time + time * noise
    C_pred, noise_pred = model_fn(x_noisy, t)
    x_rec = x_noisy - C_pred * time - time * noise_pred
    if weighting_loss:
        w1, w2 = ((t - 1) / t) ** 2 + 1, (t / (1 - t + eps)) ** 2 + 1
    else:
        w1 = w2 = torch.ones_like(t)
    sse = lambda a, b: ((a - b) ** 2).sum(dim=[1, 2, 3])
    loss_simple = w1 * sse(C_pred, C) + w2 * sse(noise_pred, noise)
    B, n = C.shape[0], C[0].numel()
    # NB reference quirk kept on purpose: rec_weight is reshaped to [B, 1] (ddm_const_2.py:566) and multiplied with the
    # [B] vector of per-sample L1 sums (:568), which BROADCASTS to a [B, B] outer product; its .sum() is therefore
    # (sum_b L1_b) * (sum_b' -log t_b' / 2): every sample's L1 term is weighted by the batch-sum of the rec weights.
    rec_weight = -torch.log(t.reshape(B, 1)) / 2
    loss_vlb = (x_rec - z0).abs().sum([1, 2, 3]) * rec_weight
    loss = loss_simple.sum() / B + loss_vlb.sum() / B
    log = {"train/loss_simple": loss_simple.detach().sum() / B / n,
           "train/loss_vlb": loss_vlb.detach().sum() / B / n,
           "train/loss": loss.detach() / B / n}
    return loss, log, (x_noisy, C_pred, noise_pred)


def latent_sample_fn_d(model_fn: Callable, x_T: Tensor, n: int, sigma_max: float = 1.0):
    """ddm_const_2.py:677-737 as called from sample(): unnormalize=False, fp64 state, no clamps.  x_T = unit normal."""
    i = torch.arange(n, dtype=torch.float64)
    ts = sigma_max + i / (n - 1) * (1.0 / n - sigma_max)
    ts = torch.cat([ts, torch.zeros(1, dtype=torch.float64)])
    x = x_T.to(torch.float64) * ts[0]
    for t_cur, t_next in zip(ts[:-1], ts[1:]):
        C, e = model_fn(x, t_cur)
        C, e = C.to(torch.float64), e.to(torch.float64)
        x0 = x - C * t_cur - e * t_cur
        x = x0 + t_next * C + t_next * e
    return x


def latent_sample_fn_s(model_fn: Callable, x_T: Tensor, epsilons: Sequence[Tensor], n: int, eps: float,
                       denoise: bool = True):
    """ddm_const_2.py:624-675 as called from sample(): uniform steps 1/n, with `denoise` the last one split into
    (1/n - eps, eps); C re-derived from the predicted x0 each step; fp32 state like the reference."""
    step = 1.0 / n
    steps = torch.full((n,), step, dtype=torch.float32)
    if denoise:
        steps = torch.cat([steps[:-1], torch.tensor([steps[-1] - eps]), torch.tensor([eps])])
    img = x_T.to(torch.float32)
    B = img.shape[0]
    cur = torch.ones((B,))
    for k, ts in enumerate(steps):
        s = torch.full((B,), float(ts))
        if k == len(steps) - 1:
            s = cur
        C, e = model_fn(img, cur)
        time, sb = _bc(cur, img), _bc(s, img)
        x0 = img - C * time - time * e
        C = -1 * x0
        mean = img - C * sb - (2 * sb * time - sb ** 2) / time * e
        sigma = torch.sqrt(2 * sb * time - sb ** 2) * (time - sb) / time
        img = mean + sigma * epsilons[k].to(mean.dtype)
        cur = cur - s
    return img


def latent_sample(sd_ae, cfg_ae, z: Tensor, scale_factor: float):
    """tail of LatentDiffusion.sample (ddm_const_2.py:611-622): un-scale, decode, (x+1)/2, clamp to [0,1]."""
    z = (1.0 / scale_factor) * z
    x = decode(sd_ae, cfg_ae, z.to(torch.float32))
    return torch.clamp((x + 1) * 0.5, 0.0, 1.0)
