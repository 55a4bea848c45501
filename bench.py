#!/usr/bin/env python3
"""Benchmark of the DDM hot path on MI355X: train images/sec (+ 10-step sample images/sec).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by
torch.distributed.run, one rank per GPU over RCCL.  One "step" = one optimizer step of the full
CIFAR-10 recipe network (BASELINE.json configs[1]: uncond two-decoder UNet, 216 M params, fp32,
128 images per GPU, ddm_const schedule, dropout 0.1): q_sample -> UNet forward -> weighted loss ->
backward -> [bucketed gradient all-reduce] -> clip 1.0 + AdamW + EMA.  Weak scaling: per-GPU batch
is fixed.  After the timed steps rank 0 also times `sample(batch_size=128)` (10 NFE), measures the
implicit-GEMM kernel's average duration with HIP events (roofline object) and times the CPU oracle
on a bounded sample (cpu_baseline object).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: ~2.5 PFLOP/s dense bf16


def log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.time()


def host_cores():
    """CPU share actually usable by this process (the GPU box exposes far more cores than its cgroup grants)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sample", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the grad-accum-2 and bf16-mode side measurements of the headline line")
    ap.add_argument("--profile-only", action="store_true", help="run warmup+steps only (for rocprofv3)")
    ap.add_argument("--small", action="store_true", help="reduced-width model (debug only; result marked invalid)")
    ap.add_argument("--config", choices=["cifar", "latent", "latent-ae", "sr"], default="cifar",
                    help="cifar = BASELINE configs[1] (default, the headline metric); latent = the UNet of configs[3] alone "
                         "(uncond_unet_sd_2 on 64x64x3 latents, model_channels 128; default --batch 32); latent-ae = all of "
                         "configs[3]: 256x256 images -> frozen KL-f4 autoencoder encode -> LatentDiffusion step; sampling "
                         "ends with the decode to 256x256; sr = BASELINE configs[4]: DIV2K 4x super-resolution, 512x512 images -> "
                         "frozen KL-f4 encode -> conditional LatentDiffusion (ddm_const, use_l1) on 128x128x3 latents with "
                         "unet.cond_unet_sd.Unet (dim 128) and INJECTED condition-encoder features (the Swin-B backbone is not "
                         "part of this build); sample = 5-step latent sampler + decode to 512x512; default --batch 16")
    ap.add_argument("--augment", action="store_true",
                    help="use_augment: True as in the reference's CIFAR YAML: AugmentPipe on x_start + 9 augment labels "
                         "(SURVEY 8d asks for this as a second number; the headline line keeps it off)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="contraction precision of conv/Linear: f32 = BASELINE configs[1] (default, exact fp32 MFMA); "
                         "bf16 = configs[2] mode (bf16 MFMA operands, fp32 accumulate/storage)")
    return ap.parse_args()


def build_sr_model(dev, small=False):
    """configs/super-resolution/div2k_cond_ddm_const_ldm.yaml: KL-f4 first stage (ch 128) + cond_unet_sd.Unet(dim 128)."""
    import warnings
    from adm_amd.ddm.ddm_const import LatentDiffusion
    from adm_amd.ddm.encoder_decoder import AutoencoderKL
    from adm_amd.unet.cond_unet_sd import Unet
    torch.manual_seed(1234)
    dim = 32 if small else 128
    dd = dict(double_z=True, z_channels=3, resolution=[512, 512], in_channels=3, out_ch=3, ch=32 if small else 128, ch_mult=[1, 2, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    ae = AutoencoderKL(dd, dict(disc_start=50001, kl_weight=1e-6, disc_weight=0.5), 3)
    unet = Unet(dim=dim, dim_mults=(1, 2, 4, 4), cond_dim=dim, cond_dim_mults=(), channels=3, cond_in_dim=3,
                window_sizes1=[[8, 8], [4, 4], [2, 2], [1, 1]], window_sizes2=[[4, 4], [2, 2], [1, 1], [1, 1]], fourier_scale=16,
                cfg={"cond_net": "swin", "cond_pe": False})
    mcfg = dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True, use_augment=False, ldm=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dpm = LatentDiffusion(auto_encoder=ae, scale_factor=0.195, scale_by_std=True, default_scale=True, model=unet,
                              image_size=[512, 512], sampling_timesteps=5, loss_type="l2", start_dist="normal",
                              perceptual_weight=0.0, use_l1=True, cfg=mcfg)
    return dpm.to(dev)


def build_model(dev, small=False, config="cifar", augment=False):
    if config == "sr":
        return build_sr_model(dev, small)
    from adm_amd.ddm.ddm_const import DDPM
    from adm_amd.unet.uncond_unet import EDMPrecond
    kw = dict(model_channels=192, channel_mult=[1, 2, 2, 2], channel_mult_emb=4, num_blocks=3, attn_resolutions=[16, 8],
              dropout=0.1, label_dropout=0, augment_dim=9)
    res = 32
    if config in ("latent", "latent-ae"):      # celeb_uncond_ddm_const2_unet_ldm.yaml:42-55 (the reference's released-weights family)
        from adm_amd.ddm.ddm_const_2 import DDPM
        from adm_amd.unet.uncond_unet_sd_2 import EDMPrecond
        kw.update(model_channels=128, augment_dim=0)
        res = 64
    if small:
        kw.update(model_channels=64, num_blocks=1)
    torch.manual_seed(1234)
    unet = EDMPrecond(img_resolution=res, img_channels=3, sigma_data=1.0, model_type="DhariwalUNet", **kw)
    # the reference zero-initialises conv1 / proj (dead branches at step 0); give them the Dhariwal init
    # so the benchmark exercises every kernel with non-trivial data, as a mid-training model would
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if (name.endswith("conv1.weight") or name.endswith("proj.weight")) and float(p.abs().max()) == 0:
                fan_in = p[0].numel()
                p.copy_((torch.rand_like(p) * 2 - 1) * (1.0 / fan_in) ** 0.5)
    mcfg = dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True, use_augment=bool(augment), ldm=False)
    if config in ("latent", "latent-ae"):
        mcfg.update(eps=1e-3, sigma_min=0.001)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if config == "latent-ae":     # celeb_uncond_ddm_const2_unet_ldm.yaml:1-40: KL-f4 first stage, default scale 0.165
            from adm_amd.ddm.ddm_const_2 import LatentDiffusion
            from adm_amd.ddm.encoder_decoder import AutoencoderKL
            dd = dict(double_z=True, z_channels=3, resolution=[256, 256], in_channels=3, out_ch=3, ch=32 if small else 128,
                      ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
            ae = AutoencoderKL(dd, dict(disc_start=20001, kl_weight=1e-6, disc_weight=0.5), 3)
            mcfg.update(ldm=True, use_disloss=False)
            dpm = LatentDiffusion(auto_encoder=ae, scale_factor=0.165, scale_by_std=True, default_scale=True, model=unet,
                                  image_size=[256, 256], sampling_timesteps=10, loss_type="l2", start_dist="normal",
                                  perceptual_weight=0.0, use_l1=False, cfg=mcfg)
        else:
            dpm = DDPM(model=unet, image_size=[res, res], sampling_timesteps=10, loss_type="l2", start_dist="normal",
                       perceptual_weight=0.0, use_l1=False, cfg=mcfg)
    return dpm.to(dev)


def cpu_baseline(seconds_cap=60.0):
    """Oracle (plain PyTorch CPU restatement, kind='port') on BASELINE configs[0]: full 216 M model,
    bs=8, fp32: 1 warm-up + 2 timed train steps (fwd+bwd, loss_simple) and one 10-step sample(8)."""
    from oracle import ddm_ref, fill, unet_ref
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {cores} threads")
    cfg = unet_ref.default_cfg(variant="uncond_unet", dropout=0.1)
    sd = fill.filled_state_dict(unet_ref.param_shapes(cfg))
    sd = {k: v.requires_grad_("resample" not in k) for k, v in sd.items()}
    o = ddm_ref.OracleDDPM(sd, cfg, "const", eps=1e-4, sigma_min=0.01, sigma_max=1.0)
    o.training = True
    g = torch.Generator().manual_seed(0)
    batch = {"image": torch.rand(8, 3, 32, 32, generator=g) * 2 - 1}
    t = torch.rand(8, generator=torch.Generator().manual_seed(1)) * (1 - 1e-4) + 1e-4
    noise = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    times = []
    for i in range(3):
        t0 = time.time()
        loss, _ = o.training_step(batch, t=t, noise=noise)
        loss.backward()
        for v in sd.values():
            v.grad = None
        times.append(time.time() - t0)
        log(f"cpu_baseline: train step {i} took {times[-1]:.2f}s")
        if sum(times) > seconds_cap:
            break
    train_s = sum(times[1:]) / max(1, len(times) - 1) if len(times) > 1 else times[0]
    o.training = False
    t0 = time.time()
    o.sample(batch_size=8, x_T=torch.randn(8, 3, 32, 32, dtype=torch.float64, generator=torch.Generator().manual_seed(3)))
    sample_s = time.time() - t0
    log(f"cpu_baseline: sample(8) took {sample_s:.2f}s")
    return {"value": round(8 / train_s, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (plain PyTorch CPU) on the full 216M-param model, bs=8 fp32: {len(times) - 1 or 1} timed "
                      f"train steps (fwd+bwd) after 1 warm-up = {train_s:.2f} s/step; one 10-step sample(8) = {sample_s:.2f} s",
            "sample_images_per_sec": round(8 / sample_s, 3)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("ADM_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = bool(os.environ.get("ADM_FORCE_DIST"))       # exercise the RCCL path on a 1-GPU box
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # "nccl" IS RCCL on ROCm.  ADM_DIST_BACKEND=gloo exists for tests that rehearse N ranks on ONE card (two RCCL
        # ranks cannot share a device); ADM_LOCAL_DEVICE pins every rank to that card.
        backend = os.environ.get("ADM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from adm_amd import hip, ops
    from adm_amd.optim import BucketedGradReducer, FlatParams, FusedAdamWEMA, ema_decay_at, lr_lambda
    hip.lib()
    ops.set_compute_precision(args.dtype)

    dpm = build_model(dev, args.small, args.config, args.augment)
    dpm.train()
    flat = FlatParams(dpm)
    if use_dist:    # identical start on every rank
        dist.broadcast(flat.flat, src=0)
    reducer = BucketedGradReducer(flat, force=force_dist)
    log(f"world={world} rank={rank} buckets={len(reducer.buckets)} reducer_active={reducer.active}")
    opt = FusedAdamWEMA(flat, lr=1e-4, weight_decay=1e-4, max_norm=1.0, ema=(rank == 0))
    B = args.batch if (args.config == "cifar" or args.batch != 128) else (16 if args.config == "sr" else 32)
    R = {"cifar": 32, "latent": 64, "latent-ae": 256, "sr": 512}[args.config]
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    batches = [{"image": torch.rand(B, 3, R, R, device=dev, generator=gen) * 2 - 1} for _ in range(2)]
    if args.config == "sr":       # the condition encoder's outputs for a 128x128 low-resolution image (Swin-B geometry), synthetic
        for bt in batches:
            bt["cond"] = [torch.randn(B, 128 << i, 32 >> i, 32 >> i, device=dev, generator=gen) for i in range(4)]

    def train_step(it):
        flat.zero_grad()
        loss, _ = dpm.training_step(batches[it & 1])
        loss.backward()
        reducer.finish()
        # EMA every 8th step as in the recipe (ema_update_every: 8); decay from the reference's warm-up law,
        # evaluated late in training (step 400k) so the lerp is a real lerp, not the early-training copy
        opt.step(lr=1e-4 * lr_lambda(400000 + it, 1e-4, 5e-6, 800000), grad_scale=1.0 / world,
                 ema_decay=ema_decay_at(400000 + it) if (it % 8 == 0) else None)
        return loss

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    log("model built; warm-up")
    for i in range(args.warmup):
        train_step(i)
    barrier()
    uploads0 = ops.table_uploads
    log("warm-up done; timing")
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = train_step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    if args.profile_only:
        if use_dist:
            dist.destroy_process_group()
        return
    ms_per_step = dt / args.steps * 1e3
    log(f"gradient-table uploads (pinned, asynchronous): {uploads0} in the warm-up, {ops.table_uploads - uploads0} in the timed steps")
    log(f"{args.steps} steps in {dt:.3f}s = {ms_per_step:.1f} ms/step")
    value = world * B * args.steps / dt
    final_loss = float(loss.detach())

    # ---- roofline of the dominant kernel: the fp32-MFMA implicit GEMM (fwd + dgrad launches) and wgrad ----
    roof = None
    sample_ips = None
    # one more step with per-launch HIP events.  EVERY rank runs it (the step contains the gradient all-reduces: a step
    # on rank 0 alone would wait for its peers forever); only rank 0 records and reports.
    # The profiled step runs on ONE stream (the weight gradients' side stream is off under PROFILE; the second decoder's
    # stream is switched off here): a kernel that shares the CUs with a kernel of another stream reads longer than it is.
    if rank == 0:
        ops.PROFILE = []
    branch, ops.BRANCH_STREAM = ops.BRANCH_STREAM, False
    train_step(args.warmup + args.steps)
    ops.BRANCH_STREAM = branch
    barrier()
    if reducer.active:        # two more untimed steps that measure how much of the all-reduce is NOT hidden under backward
        reducer.measure = True
        for i in range(2):
            train_step(args.warmup + args.steps + 1 + i)
        barrier()
        reducer.measure = False
    if rank == 0:
        recs, ops.PROFILE = ops.PROFILE, None
        log(f"profiled step: {len(recs)} GEMM-shaped launches")
        by = {}
        shapes = {}
        for kind, flops, e0, e1, tag in recs:
            ms_ = e0.elapsed_time(e1)
            a = by.setdefault(kind, [0.0, 0.0, 0])
            a[0] += flops; a[1] += ms_; a[2] += 1
            sa = shapes.setdefault(tag, [0.0, 0.0, 0])
            sa[0] += flops; sa[1] += ms_; sa[2] += 1
        if os.environ.get("ADM_BENCH_SHAPES"):
            for tag, (f_, m_, n_) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                log(f"  {tag:44s} n={n_:3d} total={m_:8.2f} ms  {f_ / max(m_, 1e-9) / 1e9:7.1f} TFLOP/s")
        # Dominant kernel: the Winograd F(2,3) 3x3-conv kernel when it ran (fp32), else the direct implicit GEMM.
        # Convention of every entry below: `achieved` / `frac` count the flops the MFMA pipe EXECUTED -- the
        # Winograd kernels execute 2/3 of the direct convolution's 2*M*N*9*Cin -- so frac = achieved / peak is a true roofline
        # fraction (<= 1) and is comparable with the PMC MFMA-busy fraction; the ALGORITHMIC (direct-convolution) rate, which
        # is what images/s follow, is carried beside it as `algorithmic` / `algorithmic_over_peak` (may exceed 1).
        dom_kind = max((k for k in ("wino2x6", "wino2h3", "wino2", "wino", "igemm") if k in by), key=lambda k: by[k][1])
        # HBM-side traffic per launch comes from rocprofv3 PMC passes of this same command (separate FETCH_SIZE /
        # WRITE_SIZE runs, FETCH doubled per MI355X_MICROARCH.md): counters cannot be read from inside the process.
        traffic_src, pmc_all = None, {}
        try:
            pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_summary.json"))[-1]
            pmc_all = json.load(open(os.path.join(ROOT, "profiles", pmc)))
            traffic_src = "profiles/" + pmc
        except Exception:
            pass
        if args.dtype != "f32" or args.config != "cifar":
            pmc_all, traffic_src = {}, None    # the committed PMC passes are of the fp32 CIFAR run
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        # wino2x6: f32 products carried by SIX bf16 MFMAs (exact three-term split): its MFMA pipe executes 6 x 4/9 of the algorithmic
        # flops as bf16 flops and is priced against the bf16 peak
        PEAKS = {"attnh3": PEAK_BF16_MFMA_TFLOPS, "gemmh3": PEAK_BF16_MFMA_TFLOPS, "wgrad_wino2h3": PEAK_BF16_MFMA_TFLOPS, "wgrad_gemmh3": PEAK_BF16_MFMA_TFLOPS, "wino2h3": PEAK_BF16_MFMA_TFLOPS, "wino2x6": PEAK_BF16_MFMA_TFLOPS, "wgrad_wino2x6": PEAK_BF16_MFMA_TFLOPS, "gemmx6": PEAK_BF16_MFMA_TFLOPS, "wgrad_gemmx6": PEAK_BF16_MFMA_TFLOPS}
        PMC_CLASS = {"attnh3": "attn", "gemmh3": "gemm_x6", "wgrad_wino2h3": "wgrad_x6", "wgrad_gemmh3": "wgrad_x6", "wino2h3": "wino2d_x6", "wino2x6": "wino2d_x6", "wino2": "wino2d", "wgrad_wino2": "wgrad_wino2d", "wgrad_wino2x6": "wgrad_x6", "gemmx6": "gemm_x6", "wgrad_gemmx6": "wgrad_x6", "attn": "attn"}
        EXEC = {"attnh3": 3.0, "gemmh3": 3.0, "wgrad_wino2h3": 3.0 * 4.0 / 9.0, "wgrad_gemmh3": 3.0, "wino2h3": 3.0 * 4.0 / 9.0, "wino2x6": 6.0 * 4.0 / 9.0, "wgrad_wino2x6": 6.0 * 4.0 / 9.0, "gemmx6": 6.0, "wgrad_gemmx6": 6.0, "wino2": 4.0 / 9.0, "wino": 2.0 / 3.0, "wgrad_wino2": 4.0 / 9.0, "wgrad_wino": 2.0 / 3.0, "attn": 1.0, "igemm": 1.0, "wgrad": 1.0}
        NAMES = {"wino2h3": "wino2d_x6_kernel<1> (3x3 conv forward, 2-D Winograd F(2x2,3x3); f32 products as THREE fp16 MFMAs on a two-term "
                            "round-to-nearest fp16 split scaled by the operand's max (written by the GroupNorm kernel), f32 accumulate)",
                 "wino2x6": "wino2d_x6_kernel<0> (3x3 conv forward + data-gradient, 2-D Winograd F(2x2,3x3); f32 products as six bf16 MFMAs "
                            "on the exact three-term bf16 split, f32 accumulate)",
                 "wino2": "igemm_wino2d_kernel (3x3 conv forward + data-gradient, 2-D Winograd F(2x2,3x3), fp32 MFMA)",
                 "wino": "igemm_wino_kernel (3x3 conv forward + data-gradient, Winograd F(2,3): fused-upsample / odd-height layers)",
                 "igemm": ("igemm_f32_kernel" if args.dtype == "f32" else "igemm_bf16_kernel") +
                          " (1x1 / Linear / small-map / fused-upsample convs: forward + data-gradient)",
                 "wgrad_gemmx6": "wgrad_x6_kernel<1> (1x1 weight gradients with >= 8192 pixels; f32 products as six bf16 MFMAs)",
                 "gemmh3": "gemm_x6_kernel<1> (1x1 convs with >= 8192 pixels whose input came with a bound, forward + data-gradient; three fp16 MFMAs per f32 product)",
                 "gemmx6": "gemm_x6_kernel (1x1 convs with >= 8192 pixels, forward + data-gradient; f32 products as six bf16 MFMAs)",
                 "wgrad_wino2x6": "wgrad_x6_kernel (3x3 weight gradient, 2-D Winograd F(3x3,2x2); f32 products as six bf16 MFMAs on the "
                                  "exact three-term bf16 split of both operands, f32 accumulate)",
                 "wgrad_wino2h3": "wgrad_x6_kernel<0, false, 1> (3x3 weight gradient, 2-D Winograd F(3x3,2x2); f32 products as three fp16 MFMAs "
                                  "on the two-term fp16 split of both operands, scaled by the bounds their producers wrote; f32 accumulate)",
                 "wgrad_gemmh3": "wgrad_x6_kernel<1, false, 1> (1x1 weight gradients, three fp16 MFMAs per f32 product)",
                 "wgrad_wino2": "wgrad_wino_kernel<2> (3x3 weight gradient, 2-D Winograd F(3x3,2x2))",
                 "wgrad_wino": "wgrad_wino_kernel (3x3 weight gradient, Winograd F(3,2): fused-upsample layers)",
                 "wgrad": "wgrad_f32_kernel (direct: 1x1 / Linear / small-map layers)",
                 "attnh3": "attn_fwd_h3_kernel (attention forward, both products as three fp16 MFMAs per f32 product; bound of qkv from the qkv conv's epilogue)",
                 "attn": "attn_fwd / attn_bwd_dq / attn_bwd_dkv"}

        def mfma_entry(kind):
            fl, ms, n = by[kind]
            ex = EXEC[kind]
            pk = PEAKS.get(kind, peak)
            pc = PMC_CLASS.get(kind, kind)
            e = {"kernel": NAMES[kind], "bound": "mfma", "achieved": round(fl * ex / ms / 1e9, 2), "peak": pk,
                 "unit": "TFLOP/s", "frac": round(fl * ex / ms / 1e9 / pk, 4),
                 "algorithmic": round(fl / ms / 1e9, 2), "algorithmic_over_peak": round(fl / ms / 1e9 / pk, 4),
                 "executed_over_algorithmic_flops": round(ex, 4),
                 "algorithmic_flops_per_launch": round(fl / max(n, 1)), "launches_per_step": n,
                 "avg_launch_ms": round(ms / max(n, 1), 4), "ms_per_step_in_kernel": round(ms, 2),
                 "traffic": pmc_all.get(pc, {}).get("traffic_bytes_per_launch"),
                 "mfma_busy_pmc": pmc_all.get(pc, {}).get("mfma_busy_fraction"),
                 "sustained_clock_GHz_pmc": pmc_all.get(pc, {}).get("effective_clock_GHz")}
            if kind in ("wgrad_wino2x6", "wgrad_wino2h3"):
                e["f32_equivalent"] = round(fl * 4.0 / 9.0 / ms / 1e9, 2)
            if kind in ("gemmx6", "gemmh3", "wgrad_gemmx6", "wgrad_gemmh3"):
                e["f32_equivalent"] = round(fl / ms / 1e9, 2)
            if kind == "wino2h3":
                e["f32_equivalent"] = round(fl * 4.0 / 9.0 / ms / 1e9, 2)
                e["f32_equivalent_over_f32_mfma_peak"] = round(fl * 4.0 / 9.0 / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4)
                e["note"] = ("three fp16 MFMAs per f32 product (two-term round-to-nearest split after a power-of-two scaling by the operand's "
                             "maximum): `achieved` = executed fp16 MFMA flops / kernel time against the dense 16-bit MFMA peak; error vs fp64 = "
                             "the six-bf16 form's (tests/test_hip_ops.py::test_conv_h3_error_vs_fp64).  Halving the executed flops per product "
                             "LOWERED this fraction (round 2: 0.29 with six products) while the kernel got 1.46x faster: `f32_equivalent` = the "
                             "f32 products per second it delivers, `f32_equivalent_over_f32_mfma_peak` = that against the f32 MFMA pipe it "
                             "replaces.  The kernel is bound by the path from the L2 into the CU, not by the matrix pipe: a stage of a 96-cout "
                             "workgroup fetches 56 KB in ~2400 cycles = 23-24 B/clk per CU against ~30 B/clk deliverable with every CU asking "
                             "(DESIGN.md section 4, 'The fetch bound'; workgroup forms are chosen by bytes fetched per MFMA)")
            if kind == "wino2x6":
                e["f32_equivalent"] = round(fl * 4.0 / 9.0 / ms / 1e9, 2)
                e["note"] = ("`achieved` = executed bf16 MFMA flops (six bf16 products per f32 product x 4/9 of the direct convolution's "
                             "flops) / kernel time, priced against the dense bf16 MFMA peak; `f32_equivalent` = the f32 products per "
                             "second this replaces (the f32-MFMA kernel of conv_wino2d.hip reaches 91 of its 157.3 TFLOP/s peak on the "
                             "same layers: ADM_BF16X6=0).  Results are f32-accurate: error vs fp64 at or below the f32 MFMA kernel's "
                             "(tests/test_hip_ops.py::test_conv_x6_error_vs_fp64).  The kernel is bound by its data path (LDS "
                             "fragment traffic, operand transform + split), not by the matrix pipe: DESIGN.md section 4.")
            if kind == "attn":
                e["note"] = ("algorithmic flops: 4 L^2 d forward, 10 L^2 d backward per (image, head); the backward EXECUTES "
                             "7 products for these 5, which mfma_busy_pmc sees and `achieved` does not count")
            return e

        roof = mfma_entry(dom_kind)
        roof.update({"traffic_unit": "bytes/launch (L2-miss side: HBM + Infinity Cache), rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE",
                     "traffic_source": traffic_src,
                     "convention": "achieved = EXECUTED MFMA flops / kernel time (HIP events around every launch of one profiled step); "
                             "frac = achieved / peak.  Winograd executes 4/9 (2-D F(2x2,3x3)) or 2/3 (1-D F(2,3)) of the direct convolution's flops: "
                             "`algorithmic` is the direct-convolution rate (SURVEY 8d's 213.9 GFLOP/image figures)."})
        for kind, key in (("wino2x6", "wino2d_x6"), ("wino2h3", "wino2d_h3"), ("wino2", "wino2d_f32"), ("wino", "wino_1d"), ("igemm", "igemm_direct"), ("gemmx6", "gemm_x6"), ("gemmh3", "gemm_h3"), ("wgrad_wino2x6", "wgrad_x6"), ("wgrad_wino2h3", "wgrad_h3"), ("wgrad_gemmx6", "wgrad_gemm_x6"), ("wgrad_gemmh3", "wgrad_gemm_h3"), ("wgrad_wino2", "wgrad_wino2d"), ("wgrad_wino", "wgrad_wino"), ("wgrad", "wgrad"), ("attn", "attention"), ("attnh3", "attention_fwd_h3")):
            if kind in by and kind != dom_kind:
                roof[key] = mfma_entry(kind)
        # whole step against the MFMA roof: every GEMM-shaped launch of the profiled step
        alg = sum(by[k][0] for k in EXEC if k in by)
        exe = sum(by[k][0] * EXEC[k] for k in EXEC if k in by)
        at_peak_ms = sum(by[k][0] * EXEC[k] / PEAKS.get(k, peak) / 1e9 for k in EXEC if k in by)   # time the MFMA pipe needs at its peak
        roof["step"] = {"algorithmic_tflop": round(alg / 1e12, 3), "executed_tflop": round(exe / 1e12, 3),
                        "ms_per_step": round(ms_per_step, 2),
                        "algorithmic_tflops": round(alg / ms_per_step / 1e9, 2), "executed_tflops": round(exe / ms_per_step / 1e9, 2),
                        "mfma_ms_at_peak": round(at_peak_ms, 2),
                        "frac": round(at_peak_ms / ms_per_step, 4),
                        "ms_in_mfma_kernels": round(sum(by[k][1] for k in EXEC if k in by), 2),
                        "note": "frac = (executed MFMA flops of every GEMM-shaped kernel, each divided by the peak of the MFMA type it "
                                "runs on) / wall time of a whole optimiser step: the fraction of the step the matrix pipe would be busy "
                                "at its peak rate"}
        if "gn" in by:      # the HBM-bound part of the ResBlock: GroupNorm + scale/shift + SiLU + dropout, forward and backward
            by3, ms5, n5 = by["gn"]        # `flops` slot carries MINIMAL bytes: 8 B/elem fwd (read x, write y), 12 bwd (x, dy, dx) (+4 residual)
            gn_traffic = pmc_all.get("gn", {}).get("traffic_bytes_per_launch")
            gn_launches = pmc_all.get("gn", {}).get("launches_in_pmc_run")
            roof["groupnorm"] = {"kernel": "gn_* (GroupNorm+scale/shift+SiLU+dropout fwd/bwd)", "bound": "hbm",
                                 "achieved": round(by3 / ms5 / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(by3 / ms5 / 1e6 / 8000.0, 4), "calls_per_step": n5,
                                 "minimal_bytes_per_step": round(by3), "ms_per_step_in_kernel": round(ms5, 2),
                                 "traffic": gn_traffic, "traffic_launches_in_pmc_run": gn_launches,
                                 "note": "achieved = MINIMAL bytes (x read once + y written = 8 B/element forward; x, dy read + dx "
                                         "written = 12 B/element backward, +4 where the residual gradient is added in the same "
                                         "pass) / time.  The multi-pass kernels read x (and dy) twice on the large maps: "
                                         "`traffic` is the PMC bytes per kernel launch (several launches per call)"}
    # ---- 10-step sampling, every rank samples its own batch, no collectives ----
    if not args.no_sample:
        dpm.eval()
        skw = {"cond": batches[0]["cond"]} if args.config == "sr" else {}
        dpm.sample(batch_size=min(B, 16), **({"cond": [c[:min(B, 16)] for c in skw["cond"]]} if skw else {}))
        barrier()
        t0 = time.perf_counter()
        img = dpm.sample(batch_size=B, **skw)
        barrier()
        st = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([st], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            st = float(tmax)
        assert img.shape == (B, 3, R, R) and img.dtype == (torch.float32 if args.config in ("latent-ae", "sr") else torch.float64)
        sample_ips = world * B / st
        log(f"sample({B}) took {st:.3f}s")

    # ---- side measurements carried on the ONE headline line (N = 1, configs[1] only) --------------------
    #  * gradient accumulation 2 as in the reference YAML (configs/cifar10/ddm_uncond_const_uncond_unet.yaml:52-61): one optimiser
    #    step = two micro-batches of B images (loss / 2 each), images/s = 2 B / time per optimiser step;
    #  * the opt-in bf16 contraction mode (BASELINE configs[2]'s per-GPU share): same model object, conv / Linear operands rounded to
    #    bf16, everything else as in the headline.  Never part of `value`.
    extras = None
    if world == 1 and args.config == "cifar" and args.dtype == "f32" and not args.no_extras and not args.profile_only:
        dpm.train()

        def accum2_step(it):
            flat.zero_grad()
            for ga in range(2):
                reducer.enabled = ga == 1
                loss, _ = dpm.training_step(batches[(it + ga) & 1])
                (loss / 2).backward()
            reducer.finish()
            opt.step(lr=1e-4 * lr_lambda(400000 + it, 1e-4, 5e-6, 800000), grad_scale=1.0 / world, ema_decay=None)

        def timed(fn, warm, n):
            for i in range(warm):
                fn(i)
            barrier()
            t0 = time.perf_counter()
            for i in range(n):
                fn(warm + i)
            barrier()
            return (time.perf_counter() - t0) / n

        n_x = max(2, min(6, args.steps // 2))
        t_acc = timed(accum2_step, 1, n_x)
        ops.set_compute_precision("bf16")
        try:
            t_bf = timed(train_step, 2, n_x)
        finally:
            ops.set_compute_precision("f32")
        extras = {"grad_accum_2": {"images_per_sec": round(2 * B / t_acc, 2), "ms_per_optimizer_step": round(t_acc * 1e3, 2), "steps": n_x,
                                   "note": "gradient_accumulate_every: 2 of the reference YAML: two micro-batches of B per optimiser step"},
                  "bf16_mode": {"images_per_sec": round(B / t_bf, 2), "ms_per_step": round(t_bf * 1e3, 2), "steps": n_x, "dtype": "bf16",
                                "note": "opt-in bf16 contraction mode (BASELINE configs[2] per-GPU share: bf16 MFMA operands + bf16 activation "
                                        "storage, f32 accumulate / master weights); `python bench.py --dtype bf16` is the full line"}}
        log(f"extras: accum-2 {extras['grad_accum_2']['images_per_sec']} images/s, bf16 mode {extras['bf16_mode']['images_per_sec']} images/s")

    if rank == 0:
        what = {"cifar": "CIFAR-10 32x32 uncond DDM UNet", "latent": "64x64x3-latent uncond DDM UNet (configs[3], UNet only)",
                "latent-ae": "CelebA-HQ-256-shaped latent DDM: frozen KL-f4 AE + 64x64x3-latent UNet (configs[3])",
                "sr": "DIV2K 4x SR: frozen KL-f4 AE + conditional cond_unet_sd on 128x128x3 latents (configs[4])"}[args.config]
        out = {"metric": f"train images/sec ({what}, 1 optimizer step/iter) + {5 if args.config == 'sr' else 10}-step sample images/sec",
               "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic",
               "sample_images_per_sec": None if sample_ips is None else round(sample_ips, 2),
               "config": {"workload": ("BASELINE configs[3] UNET ONLY (uncond_unet_sd_2, model_channels 128, 64x64x3 latents, "
                                       "ddm_const_2; no autoencoder in this line): " if args.config == "latent"
                                       else "BASELINE configs[3]: 256x256 images -> frozen KL-f4 AutoencoderKL.encode (55M params, no "
                                       "grad) -> LatentDiffusion (ddm_const_2, uncond_unet_sd_2 model_channels 128) training step; "
                                       "sample = 10-step latent sampler + AutoencoderKL.decode to 256x256: " if args.config == "latent-ae"
                                       else "BASELINE configs[4] (per-GPU share): 512x512 images -> frozen KL-f4 AutoencoderKL.encode -> conditional "
                                       "LatentDiffusion (ddm_const, use_l1) with unet.cond_unet_sd.Unet (dim 128, 87M params) on 128x128x3 "
                                       "latents; condition = INJECTED synthetic Swin-B-shaped feature pyramid (the backbone is not part of this "
                                       "build); sample = 5-step latent sampler + decode to 512x512: " if args.config == "sr"
                                       else ("BASELINE configs[1]" if args.dtype == "f32" else "BASELINE configs[2] (per-GPU share)") +
                                       ": CIFAR-10 32x32 uncond two-decoder DhariwalUNet (216M params), ") +
                                      f"bs={B}/GPU {'fp32' if args.dtype == 'f32' else 'bf16 MFMA operands, fp32 accumulate+storage'}, {'ddm_const' if args.config in ('cifar', 'sr') else 'ddm_const_2'} schedule, dropout 0.1, " +
                                      ("loss_simple + latent L1 term" if args.config in ("latent-ae", "sr") else "loss_simple (LPIPS term needs unfetchable VGG16 weights)") +
                                      ", clip 1.0 + AdamW + EMA(every 8)" + (", use_augment (AugmentPipe p=0.15 + labels)" if args.augment else ""),
                          "global_batch": world * B, "image": f"3x{R}x{R}", "sampling_timesteps": 5 if args.config == "sr" else 10,
                          "parallelism": f"dp{world}", "valid": not args.small},
               "final_loss": round(final_loss, 4), "roofline": roof,
               "dist": {"backend": (dist.get_backend() if use_dist else None),
                        "nranks": (dist.get_world_size() if use_dist else 1),
                        "reducer_active": reducer.active, "buckets": len(reducer.buckets), "bucket_MiB": 64,
                        "allreduce_exposed_ms_per_step": round(reducer.exposed_ms / max(1, reducer.finishes), 3),
                        "note": "exposed = host-measured time finish() spent waiting for the all-reduce side stream after the "
                                "backward's own kernels had drained (0 when the reducer is inactive)"}}
        if extras is not None:
            out["also_measured"] = extras
        if not args.no_cpu_baseline and world == 1 and args.config == "cifar":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
