"""GPU parity tests of the conditional super-resolution denoiser (BASELINE configs[4]).

Op level: every new HIP kernel (through the C ABI) against plain PyTorch fp32 on the CPU.  Block level: each new block class at
the FULL width of the DIV2K recipe against the vectors the imported reference produced (tests/golden/g15_cond_blocks.npz).
Model level: the whole single-decoder network at reduced width, eval and train mode, outputs + every parameter gradient +
BatchNorm running statistics against g14_cond_unet.npz; the two-decoder variant against the CPU oracle.
Tolerance: north_star's rtol 1e-3 / atol 1e-4 (tests/parity.py)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cond_unet_ref as R
from oracle import fill
from parity import close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def oc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from adm_amd import hip, ops_cond
    hip.lib()
    return ops_cond


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2)


def rt(shape, tag, scale=1.0):
    return fill.hash_tensor(shape, tag, scale)


# ------------------------------------------------------------------------------------------------ op level
@pytest.mark.parametrize("O,ci,k", [(64, 32, 3), (128, 256, 3), (32, 96, 1), (5, 7, 3)])
def test_weight_standardize(oc, O, ci, k):
    w = rt((O, ci, k, k), f"ws{O}{ci}", 0.3) + 0.05
    g = rt((O, ci, k, k), f"wsg{O}{ci}", 1.0)
    wr = w.clone().requires_grad_(True)
    y = R.ws_weight(wr)
    (y * g).sum().backward()
    wd = w.cuda().requires_grad_(True)
    yd = oc.weight_standardize(wd)
    close(yd, y)
    (yd * g.cuda()).sum().backward()
    close(wd.grad, wr.grad)
    with torch.no_grad():      # cached path (sampling)
        close(oc.weight_standardize(wd), y)


@pytest.mark.parametrize("B,H,C", [(2, 8, 128), (1, 16, 512), (3, 5, 32), (2, 4, 1024)])
def test_layer_norm_channels(oc, B, H, C):
    x = rt((B, C, H, H), f"ln{C}", 2.0) + 0.3
    g = 1 + rt((1, C, 1, 1), f"lng{C}", 0.2)
    gy = rt((B, C, H, H), f"lny{C}", 1.0)
    xr, gr = x.clone().requires_grad_(True), g.clone().requires_grad_(True)
    y = R.layer_norm_c(xr, gr)
    (y * gy).sum().backward()
    xd, gd = nhwc(x).requires_grad_(True), g.cuda().requires_grad_(True)
    yd = oc.layer_norm_c(xd, gd)
    close(nchw(yd), y)
    (yd * nhwc(gy)).sum().backward()
    close(nchw(xd.grad), xr.grad)
    close(gd.grad, gr.grad)


@pytest.mark.parametrize("B,H,C,training", [(2, 8, 128, True), (3, 4, 32, True), (2, 16, 512, False), (4, 2, 256, True)])
def test_batch_norm(oc, B, H, C, training):
    bn_ref = torch.nn.BatchNorm2d(C, momentum=0.03, eps=0.001)
    with torch.no_grad():
        bn_ref.weight.copy_(1 + rt((C,), f"bnw{C}", 0.2)); bn_ref.bias.copy_(rt((C,), f"bnb{C}", 0.1))
        bn_ref.running_mean.copy_(rt((C,), f"bnm{C}", 0.1)); bn_ref.running_var.copy_(1 + rt((C,), f"bnv{C}", 0.3))
    import copy
    bn_dev = copy.deepcopy(bn_ref).cuda()
    bn_ref.train(training); bn_dev.train(training)
    x = rt((B, C, H, H), f"bnx{C}", 2.0) + 0.5
    gy = rt((B, C, H, H), f"bny{C}", 1.0)
    xr = x.clone().requires_grad_(True)
    y = bn_ref(xr)
    (y * gy).sum().backward()
    xd = nhwc(x).requires_grad_(True)
    yd = oc.batch_norm(xd, bn_dev, training)
    close(nchw(yd), y)
    (yd * nhwc(gy)).sum().backward()
    close(nchw(xd.grad), xr.grad)
    close(bn_dev.weight.grad, bn_ref.weight.grad)
    close(bn_dev.bias.grad, bn_ref.bias.grad)
    close(bn_dev.running_mean, bn_ref.running_mean, rtol=1e-5, atol=1e-6)
    close(bn_dev.running_var, bn_ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_dev.num_batches_tracked) == int(bn_ref.num_batches_tracked)


@pytest.mark.parametrize("Hi,Wi,Ho,Wo,align", [(4, 4, 32, 32, True), (8, 8, 64, 64, True), (1, 1, 8, 8, True), (8, 8, 32, 32, False),
                                               (16, 12, 5, 7, True), (6, 10, 6, 10, True), (32, 32, 8, 8, False), (3, 5, 9, 4, False)])
def test_bilinear(oc, Hi, Wi, Ho, Wo, align):
    B, C = 2, 64
    x = rt((B, C, Hi, Wi), f"bl{Hi}{Wo}", 1.0)
    gy = rt((B, C, Ho, Wo), f"blg{Hi}{Wo}", 1.0)
    xr = x.clone().requires_grad_(True)
    y = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=align)
    (y * gy).sum().backward()
    xd = nhwc(x).requires_grad_(True)
    yd = oc.bilinear(xd, Ho, Wo, align)
    close(nchw(yd), y)
    (yd * nhwc(gy)).sum().backward()
    close(nchw(xd.grad), xr.grad)
    # the stem's concatenation: written behind 3 latent channels of a wider tensor
    buf = torch.zeros(B, Ho, Wo, 96, device="cuda")
    oc.bilinear_into(nhwc(x), buf, 3, align)
    close(nchw(buf)[:, 3:3 + C], y.detach())
    assert float(buf[..., :3].abs().max()) == 0 and float(buf[..., 3 + C:].abs().max()) == 0


def test_activations_fourier_pool(oc):
    x = rt((4, 8, 8, 64), "act.x", 3.0)
    for fn, ref in ((oc.gelu, F.gelu), (lambda t: oc.relu_dropout(t, 0.0), F.relu)):
        xd = x.cuda().requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        y, yr = fn(xd), ref(xr)
        close(y, yr)
        y.sum().backward(); yr.sum().backward()
        close(xd.grad, xr.grad)
    # dropout: same mask in forward and backward, scaled by 1 / keep
    torch.manual_seed(5)
    xd = x.cuda().requires_grad_(True)
    y = oc.relu_dropout(xd, 0.1)
    pos = x.cuda() > 0
    keep = (y != 0) | ~pos
    frac = ((~keep) & pos).float().sum().item() / pos.float().sum().item()      # dropped share of the (visible) positive entries
    assert abs(frac - 0.1) < 0.02, frac
    close(y[keep], (F.relu(x.cuda()) / 0.9)[keep])
    y.sum().backward()
    close(xd.grad, ((x.cuda() > 0) & keep).float() / 0.9)
    t = torch.tensor([1e-4, 0.3, 1.0]).log()
    W = rt((16,), "ff.W", 4.0)
    close(oc.fourier_features(t.cuda(), W.cuda()), R.gaussian_fourier(W, t))
    p = rt((2, 32, 16, 16), "pool", 1.0)
    close(nchw(oc.avg_pool(nhwc(p), 4)), F.avg_pool2d(p, 4))
    close(nchw(oc.avg_pool(nhwc(p), 1)), p)


@pytest.mark.parametrize("B,Lq,Lk,H,D,scale", [(2, 16, 1024, 8, 16, 1.0), (2, 16, 256, 8, 64, 1.0), (3, 1, 64, 8, 4, 1.0),
                                               (1, 256, 256, 4, 32, 32 ** -0.5), (2, 70, 130, 2, 8, 0.5), (2, 16, 1024, 8, 32, 1.0)])
def test_mha_cross_attention(oc, B, Lq, Lk, H, D, scale):
    C = H * D
    q, k, v = rt((B, Lq, C), f"mq{Lq}{D}", 1.0), rt((B, Lk, C), f"mk{Lk}{D}", 1.0), rt((B, Lk, C), f"mv{Lk}{D}", 1.0)
    g = rt((B, Lq, C), f"mg{Lq}{D}", 1.0)
    qr, kr, vr = [t.clone().requires_grad_(True) for t in (q, k, v)]
    hd = lambda t, L: t.reshape(B, L, H, D).permute(0, 2, 1, 3)
    a = ((hd(qr, Lq) * scale) @ hd(kr, Lk).transpose(-1, -2)).softmax(-1) @ hd(vr, Lk)
    o = a.permute(0, 2, 1, 3).reshape(B, Lq, C)
    (o * g).sum().backward()
    qd, kd, vd = [t.cuda().requires_grad_(True) for t in (q, k, v)]
    od = oc.mha(qd, kd, vd, H, scale)
    close(od, o)
    (od * g.cuda()).sum().backward()
    close(qd.grad, qr.grad); close(kd.grad, kr.grad); close(vd.grad, vr.grad)


def test_self_attention_packed_and_linear_attention(oc):
    B, h, w = 2, 16, 16
    qkv = rt((B, 384, h, w), "sa.qkv", 1.5)
    g = rt((B, 128, h, w), "sa.g", 1.0)
    # bottleneck Attention core (cond_unet_sd.py:544-553)
    qr = qkv.clone().requires_grad_(True)
    q, k, v = [t.reshape(B, 4, 32, h * w) for t in qr.chunk(3, dim=1)]
    sim = torch.einsum("bhdi,bhdj->bhij", q * 32 ** -0.5, k).softmax(-1)
    o = torch.einsum("bhij,bhdj->bhid", sim, v).permute(0, 1, 3, 2).reshape(B, 128, h, w)
    (o * g).sum().backward()
    qd = nhwc(qkv).reshape(B, h * w, 384).requires_grad_(True)
    od = oc.self_attention_packed(qd, 4, 32 ** -0.5)
    close(od.reshape(B, h, w, 128).permute(0, 3, 1, 2), o)
    (od * nhwc(g).reshape(B, h * w, 128)).sum().backward()
    close(qd.grad.reshape(B, h, w, 384).permute(0, 3, 1, 2), qr.grad)
    # LinearAttention core (cond_unet_sd.py:516-529) at several pixel counts (1 chunk, ragged chunks, many chunks)
    for hh, ww in ((16, 16), (24, 20), (64, 64)):
        qkv = rt((B, 384, hh, ww), f"la.qkv{hh}", 1.5)
        g = rt((B, 128, hh, ww), f"la.g{hh}", 1.0)
        qr = qkv.clone().requires_grad_(True)
        q, k, v = [t.reshape(B, 4, 32, hh * ww) for t in qr.chunk(3, dim=1)]
        ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(-1), v / (hh * ww))
        o = torch.einsum("bhde,bhdn->bhen", ctx, q.softmax(-2) * 32 ** -0.5).reshape(B, 128, hh, ww)
        (o * g).sum().backward()
        qd = nhwc(qkv).reshape(B, hh * ww, 384).requires_grad_(True)
        od = oc.linear_attention(qd)
        close(od.reshape(B, hh, ww, 128).permute(0, 3, 1, 2), o)
        (od * nhwc(g).reshape(B, hh * ww, 128)).sum().backward()
        close(qd.grad.reshape(B, hh, ww, 384).permute(0, 3, 1, 2), qr.grad)


@pytest.mark.parametrize("H", [16, 8, 32])
def test_spatial_att_any_size(oc, H):
    from adm_amd import ops
    B, C = 2, 64
    sd = {k: fill.fill_value("sa." + k, s) for k, s in {"map.weight": (1, C, 1, 1), "map.bias": (1,), "q_conv.weight": (1, 1, 1, 1),
                                                        "q_conv.bias": (1,), "k_conv.weight": (1, 1, 1, 1), "k_conv.bias": (1,)}.items()}
    sd = {"sa." + k: v.requires_grad_(True) for k, v in sd.items()}
    h = rt((B, C, H, H), f"sah{H}", 1.0).requires_grad_(True)
    xres = rt((B, C, H, H), f"sax{H}", 1.0).requires_grad_(True)
    gy = rt((B, C, H, H), f"sag{H}", 1.0)
    y = R.spatial_att(sd, "sa", h) + xres
    (y * gy).sum().backward()
    hd, xd = nhwc(h.detach()).requires_grad_(True), nhwc(xres.detach()).requires_grad_(True)
    mw, mb = sd["sa.map.weight"].detach().cuda().requires_grad_(True), sd["sa.map.bias"].detach().cuda().requires_grad_(True)
    qk = torch.stack([sd["sa." + n].detach().reshape(()) for n in ("q_conv.weight", "q_conv.bias", "k_conv.weight", "k_conv.bias")]).cuda()
    qk.requires_grad_(True)
    yd = oc.spatial_att_gate(ops.conv2d(hd, mw, mb), qk, hd, xd)
    close(nchw(yd), y)
    (yd * nhwc(gy)).sum().backward()
    close(nchw(hd.grad), h.grad); close(nchw(xd.grad), xres.grad); close(mw.grad, sd["sa.map.weight"].grad)
    want = torch.stack([sd["sa." + n].grad.reshape(()) for n in ("q_conv.weight", "q_conv.bias", "k_conv.weight", "k_conv.bias")])
    close(qk.grad, want, scale=float(want.abs().max()))


@pytest.mark.parametrize("B,ci,co,H,W,ks,stride,pad", [(2, 131, 128, 16, 16, 7, 1, 3), (1, 35, 32, 9, 12, 7, 1, 3), (2, 128, 128, 16, 16, 4, 2, 1),
                                                       (3, 32, 64, 8, 12, 4, 2, 1), (1, 64, 32, 6, 6, 4, 2, 1), (2, 32, 32, 64, 64, 4, 2, 1)])
def test_conv_generic_filters(oc, B, ci, co, H, W, ks, stride, pad):
    from adm_amd import ops
    x = rt((B, ci, H, W), f"cg{ci}{co}{ks}", 1.0)
    w = rt((co, ci, ks, ks), f"cgw{ci}{co}{ks}", 1.0 / math.sqrt(ci * ks * ks))
    b = rt((co,), f"cgb{ci}{co}", 0.5)
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    y = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    gy = rt(tuple(y.shape), f"cgy{ci}{co}{ks}", 1.0)
    (y * gy).sum().backward()
    cip, cop = ops.ceil32(ci), ops.ceil32(co)
    pad_c = lambda t, c: torch.cat([t, torch.zeros(t.shape[0], c - t.shape[1], *t.shape[2:])], 1) if c > t.shape[1] else t
    xd = nhwc(pad_c(x, cip)).requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yd = oc.conv2d_generic(xd, wd, bd, stride=stride, pad=pad)
    close(nchw(yd)[:, :co], y)
    (yd * nhwc(pad_c(gy, cop))).sum().backward()
    close(wd.grad, wr.grad); close(bd.grad, br.grad)
    close(nchw(xd.grad)[:, :ci], xr.grad)


# ------------------------------------------------------------------------------------------------ block level (full width)
def _load(mod, prefix):
    sd = {k: R.cond_fill_value(prefix + k, tuple(v.shape)) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd)
    for m in mod.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return mod.cuda()


def _check_block(g, name, y, ins, grad_param, rtol=1e-3):
    close(nchw(y).reshape(-1)[::7], g[name + ".y"])
    gw = fill.hash_tensor(tuple(nchw(y).shape), name + ".gy", 1.0)
    (y * nhwc(gw)).sum().backward()
    for i, t in enumerate(ins):
        got = t.grad
        got = nchw(got) if got.dim() == 4 else got.cpu()
        close(got.reshape(-1)[::7], g[f"{name}.dx{i}"])
    want = float(g[name + ".dparam_norm"])
    got = float(grad_param.grad.double().norm())
    assert abs(got - want) <= rtol * want, (name, got, want)


def test_new_block_classes_full_width_vs_reference_golden(oc):
    from adm_amd import ops
    from adm_amd.unet import cond_unet as C
    g = np.load(os.path.join(G, "g15_cond_blocks.npz"))
    temb = fill.hash_tensor((1, 512), "blk.temb", 1.0)
    for ci, co in ((256, 128), (128, 128)):
        blk = _load(C.ResnetBlock(ci, co, time_emb_dim=512, groups=8).eval(), "rb.")
        x = nhwc(fill.hash_tensor((1, ci, 32, 32), f"blk.rb{ci}", 1.0)).requires_grad_(True)
        te = temb.cuda().requires_grad_(True)
        y = blk(x, ops.silu(te))
        _check_block(g, f"resnet_{ci}_{co}", y, [x, te], blk.block1.proj.weight)
    la = _load(C.Residual(C.PreNorm(128, C.LinearAttention(128))).eval(), "la.")
    x = nhwc(fill.hash_tensor((1, 128, 32, 32), "blk.la", 1.0)).requires_grad_(True)
    _check_block(g, "linattn_128", la(x), [x], la.fn.fn.to_qkv.weight)
    fa = _load(C.Residual(C.PreNorm(512, C.Attention(512))).eval(), "fa.")
    x = nhwc(fill.hash_tensor((1, 512, 16, 16), "blk.fa", 1.0)).requires_grad_(True)
    _check_block(g, "attn_512", fa(x), [x], fa.fn.fn.to_qkv.weight)
    rn = _load(C.RelationNet(128, 128, nhead=8, layers=1, embed_dim=128, ffn_dim=256, window_size1=[8, 8], window_size2=[4, 4]).train(), "rn.")
    c = nhwc(fill.hash_tensor((2, 128, 32, 32), "blk.rnc", 1.0)).requires_grad_(True)
    f = nhwc(fill.hash_tensor((2, 128, 64, 64), "blk.rnf", 1.0)).requires_grad_(True)
    _check_block(g, "relation_128", rn(c, f), [c, f], rn.attentions[0].q_lin.weight)
    rn3 = _load(C.RelationNet(512, 512, nhead=8, layers=1, embed_dim=512, ffn_dim=1024, window_size1=[1, 1], window_size2=[1, 1]).eval(), "rn3.")
    c = nhwc(fill.hash_tensor((1, 512, 4, 4), "blk.rn3c", 1.0)).requires_grad_(True)
    f = nhwc(fill.hash_tensor((1, 512, 16, 16), "blk.rn3f", 1.0)).requires_grad_(True)
    _check_block(g, "relation_512", rn3(c, f), [c, f], rn3.attentions[0].v_lin.weight)
    ds = _load(C.Downsample(128, 128), "ds.")
    x = nhwc(fill.hash_tensor((1, 128, 32, 32), "blk.ds", 1.0)).requires_grad_(True)
    _check_block(g, "down_128", ds(x), [x], ds.weight)
    stem = _load(torch.nn.Sequential(torch.nn.Conv2d(131, 128, 7, padding=3), torch.nn.GroupNorm(8, 128)), "init_conv.")
    xs = fill.hash_tensor((1, 131, 32, 32), "blk.stem", 1.0)
    x = nhwc(torch.cat([xs, torch.zeros(1, 29, 32, 32)], 1)).requires_grad_(True)
    h0 = oc.conv2d_generic(x, stem[0].weight, stem[0].bias, stride=1, pad=3)
    y = ops.group_norm_act(h0, stem[1].weight, stem[1].bias, None, silu=False, groups=8, eps=1e-5)
    close(nchw(y).reshape(-1)[::7], g["stem_131_128.y"])
    gw = fill.hash_tensor((1, 128, 32, 32), "stem_131_128.gy", 1.0)
    (y * nhwc(gw)).sum().backward()
    close(nchw(x.grad)[:, :131].reshape(-1)[::7], g["stem_131_128.dx0"])
    want = float(g["stem_131_128.dparam_norm"])
    assert abs(float(stem[0].weight.grad.double().norm()) - want) <= 1e-3 * want


# ------------------------------------------------------------------------------------------------ model level
def build(two_decoders=False, dim=32):
    import importlib
    cfg = R.default_cfg(dim=dim, two_decoders=two_decoders)
    mod = importlib.import_module("unet.cond_unet" if two_decoders else "unet.cond_unet_sd")       # the reference's dotted paths
    m = mod.Unet(dim=dim, dim_mults=cfg["dim_mults"], cond_dim=dim, cond_dim_mults=(), channels=3, cond_in_dim=3,
                 window_sizes1=cfg["window_sizes1"], window_sizes2=cfg["window_sizes2"], fourier_scale=16,
                 cfg={"cond_net": "swin", "cond_pe": False})
    sd = R.filled_state_dict(cfg)
    m.load_state_dict(sd, strict=True)
    for mm in m.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
    return m.cuda(), cfg, sd


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_cond_unet_reduced_width_vs_reference_golden(oc, mode):
    g = np.load(os.path.join(G, "g14_cond_unet.npz"))
    m, cfg, sd = build()
    m.train(mode == "train")
    x = fill.hash_tensor((2, 3, 32, 32), "cond.x", 1.0)
    tt = torch.tensor([0.3, 0.85])
    hm = [h.cuda() for h in R.cond_features(2, 32, 32)]
    gx, gy = fill.hash_tensor((2, 3, 32, 32), "cond.gx", 1.0), fill.hash_tensor((2, 3, 32, 32), "cond.gy", 1.0)
    y1, y2 = m(x.cuda(), tt.cuda(), hm)
    close(y1, g[f"{mode}.x1"]); close(y2, g[f"{mode}.x2"])
    ((y1 * gx.cuda()).sum() + (y2 * gy.cuda()).sum()).backward()
    named = dict(m.named_parameters())
    gmax = float(g[f"{mode}.gradnorm_max"])
    bad = []
    for key in g.files:
        if key.startswith(f"{mode}.gradnorm."):
            name = key[len(mode) + 10:]
            p = named[name]
            if not p.requires_grad:
                continue
            want = float(g[key])
            got = float(p.grad.double().norm()) if p.grad is not None else 0.0
            if abs(got - want) > 2e-3 * want + 2e-4 * gmax:        # floor: analytically-zero gradients (biases in front of BatchNorm, k_lin.bias)
                bad.append((name, got, want))
    assert not bad, bad[:8]
    for key in g.files:
        if key.startswith(f"{mode}.grad."):
            name = key[len(mode) + 6:]
            close(named[name].grad.reshape(-1)[:4096], g[key], scale=max(float(g[f"{mode}.gradnorm.{name}"]) / 8, 1e-12))
    if mode == "train":
        msd = m.state_dict()
        close(msd["relation_layers_down.0.input_conv2.1.running_var"], g["train.bn.relation_layers_down.0.input_conv2.1.running_var"])
        close(msd["relation_layers_up.1.input_conv1.1.running_mean"], g["train.bn.relation_layers_up.1.input_conv1.1.running_mean"])


def test_two_decoder_variant_vs_oracle(oc):
    """unet.cond_unet.Unet (two decoders, the class the DIV2K YAML names): forward and all gradients against the CPU oracle."""
    m, cfg, sd = build(two_decoders=True)
    m.eval()
    x = fill.hash_tensor((2, 3, 32, 32), "cond.x", 1.0)
    tt = torch.tensor([0.3, 0.85])
    hm = R.cond_features(2, 32, 32)
    gx, gy = fill.hash_tensor((2, 3, 32, 32), "cond.gx", 1.0), fill.hash_tensor((2, 3, 32, 32), "cond.gy", 1.0)
    y1, y2 = m(x.cuda(), tt.cuda(), [h.cuda() for h in hm])
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "time_mlp.0.W" else v.clone())
           for k, v in sd.items()}
    o1, o2 = R.unet_forward(sdo, cfg, x, tt, hm)
    close(y1, o1.detach()); close(y2, o2.detach())
    ((y1 * gx.cuda()).sum() + (y2 * gy.cuda()).sum()).backward()
    ((o1 * gx).sum() + (o2 * gy).sum()).backward()
    gmax = max(float(v.grad.double().norm()) for v in sdo.values() if v.requires_grad and v.grad is not None)
    bad = []
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        want = sdo[name].grad
        err = float((p.grad.cpu().double() - want.double()).norm() / (want.double().norm() + 1e-4 * gmax))
        if err > 2e-3:
            bad.append((name, err))
    assert not bad, bad[:10]


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_two_decoder_variant_vs_reference_golden(oc, mode):
    """... and against the REFERENCE's own vectors (g16: tools/make_golden_cond2.py imports /root/reference/unet/cond_unet.py):
    both outputs, every parameter's gradient norm, eight sampled gradients incl. the second decoder's; the registration order of
    the parameters (= the index order of a reference optimiser state) is the reference's."""
    g = np.load(os.path.join(G, "g16_cond_unet_two_decoders.npz"))
    m, cfg, sd = build(two_decoders=True)
    assert [n for n, _ in m.state_dict().items()] == list(sd)
    m.train(mode == "train")
    x = fill.hash_tensor((2, 3, 32, 32), "cond.x", 1.0)
    tt = torch.tensor([0.3, 0.85])
    hm = [h.cuda() for h in R.cond_features(2, 32, 32)]
    gx, gy = fill.hash_tensor((2, 3, 32, 32), "cond.gx", 1.0), fill.hash_tensor((2, 3, 32, 32), "cond.gy", 1.0)
    y1, y2 = m(x.cuda(), tt.cuda(), hm)
    close(y1, g[f"{mode}.x1"]); close(y2, g[f"{mode}.x2"])
    ((y1 * gx.cuda()).sum() + (y2 * gy.cuda()).sum()).backward()
    named = dict(m.named_parameters())
    gmax = float(g[f"{mode}.gradnorm_max"])
    bad = []
    for key in g.files:
        if key.startswith(f"{mode}.gradnorm."):
            p = named[key[len(mode) + 10:]]
            if not p.requires_grad:
                continue
            want = float(g[key])
            got = float(p.grad.double().norm()) if p.grad is not None else 0.0
            if abs(got - want) > 2e-3 * want + 2e-4 * gmax:
                bad.append((key, got, want))
    assert not bad, bad[:8]
    for key in g.files:
        if key.startswith(f"{mode}.grad."):
            name = key[len(mode) + 6:]
            close(named[name].grad.reshape(-1)[:4096], g[key], scale=float(g[f"{mode}.gradnorm.{name}"]) / 8)


def test_fp64_state_and_scalar_time(oc):
    """Sampling call pattern: fp64 latent, 0-dim fp64 time, batch > 1."""
    m, cfg, sd = build()
    m.eval()
    x = fill.hash_tensor((2, 3, 32, 32), "cond.x", 1.0)
    hm = [h.cuda() for h in R.cond_features(2, 32, 32)]
    with torch.no_grad():
        a1, a2 = m(x.double().cuda(), torch.tensor(0.37, dtype=torch.float64, device="cuda"), hm)
        b1, b2 = m(x.cuda(), torch.tensor([0.37, 0.37], device="cuda"), hm)
    assert a1.dtype == torch.float32
    close(a1, b1, rtol=1e-5, atol=1e-5); close(a2, b2, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError, match="condition encoder"):
        m(x.cuda(), torch.tensor([0.3, 0.3], device="cuda"), torch.zeros(2, 3, 32, 32, device="cuda"))
