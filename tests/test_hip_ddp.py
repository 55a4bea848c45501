"""GPU, world_size = 2 on ONE card (gloo transport, since two RCCL ranks cannot share a device): the data-parallel
training step on the HIP path.  Two ranks, each with half of a batch, must end up with the parameters of one rank
stepping on the whole batch -- this drives the real pieces together: HIP backward kernels writing
into the flat gradient buffer, the direct-gradient sinks notifying the bucketed reducer, bucket all-reduces overlapped on a
side stream, and the fused clip + AdamW with grad_scale = 1/world."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fill, unet_ref

pytestmark = pytest.mark.gpu
SMALL = dict(model_channels=64, num_blocks=1, dropout=0.0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev):
    from adm_amd.ddm.ddm_const import DDPM
    from adm_amd.unet.uncond_unet import EDMPrecond
    cfg = unet_ref.default_cfg(variant="uncond_unet", **SMALL)
    kw = {k: cfg[k] for k in ("model_channels", "channel_mult", "channel_mult_emb", "num_blocks", "attn_resolutions", "dropout",
                              "augment_dim")}
    unet = EDMPrecond(img_resolution=32, img_channels=3, model_type="DhariwalUNet", **kw)
    unet.load_state_dict(fill.filled_state_dict(unet_ref.param_shapes(cfg)), strict=True)
    mcfg = dict(eps=1e-4, sigma_max=1, sigma_min=0.01, weighting_loss=True)
    return DDPM(model=unet, image_size=[32, 32], sampling_timesteps=10, perceptual_weight=0.0, cfg=mcfg).to(dev).train()


def _data(dev):
    x0 = fill.hash_tensor((8, 3, 32, 32), "ddp.x0", 1.0).to(dev)
    noise = fill.hash_tensor((8, 3, 32, 32), "ddp.noise", 1.7).to(dev)
    t = torch.tensor([0.23, 0.81, 0.5, 0.05, 0.9, 0.33, 0.64, 0.12], device=dev)
    return x0, noise, t


LR = 1e-3


def _grads(dpm, flat, red, x0, noise, t):
    """One forward + backward + reducer.finish(): returns the (reduced) flat gradient buffer."""
    flat.zero_grad()
    loss, _ = dpm.training_step({"image": x0}, t=t, noise=noise)
    loss.backward()
    red.finish()
    torch.cuda.synchronize()
    return flat.grad.detach().cpu().clone()


def _worker(rank, world, port, out_path, deterministic):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adm_amd import ops
    from adm_amd.optim import BucketedGradReducer, FlatParams, FusedAdamWEMA
    ops.DETERMINISTIC = bool(deterministic)
    dev = torch.device("cuda:0")
    dpm = _build(dev)
    flat = FlatParams(dpm)
    red = BucketedGradReducer(flat, bucket_bytes=4 << 20)
    assert red.active and len(red.buckets) >= 3
    opt = FusedAdamWEMA(flat, lr=LR, weight_decay=1e-4, max_norm=1.0, ema=False)
    x0, noise, t = _data(dev)
    sl = slice(rank * 4, (rank + 1) * 4)
    start = flat.flat.detach().cpu().clone()
    g1 = _grads(dpm, flat, red, x0[sl], noise[sl], t[sl])               # SUM over the two ranks
    opt.step(lr=LR, grad_scale=1.0 / world, ema_decay=None)
    n1 = opt.grad_norm(1.0 / world)
    torch.cuda.synchronize()
    p1 = flat.flat.detach().cpu().clone()
    g2 = _grads(dpm, flat, red, x0[sl], noise[sl], t[sl])               # second pass: buffers / tables / streams are reused
    if rank == 0:
        torch.save({"start": start, "g1": g1, "p1": p1, "n1": n1, "g2": g2, "uploads": ops.table_uploads}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def _compare_grads(got_sum, want, flat, names, world, what, tol):
    """2-rank SUM x 1/world vs the whole-batch gradient, parameter by parameter: |d| <= tol * max|g_param| elementwise (the two
    sides sum the same per-image terms in a different order).  The message names the worst parameter and its values."""
    worst = (0.0, "", 0.0, 0.0, 0.0)
    # floor of the scale: a few gradients are analytically ZERO (k_conv.bias: the softmax is shift-invariant), only rounding noise
    # of the terms that cancel is left in them -- measured against the largest gradient entry of the whole model
    floor = 1e-4 * float(want.abs().max())
    for idx, (n, o) in enumerate(zip(names, flat.offsets)):
        k = flat.params[idx].numel()
        a = got_sum[o:o + k].double() / world
        b = want[o:o + k].double()
        if float(b.abs().max()) == 0.0:
            assert float(a.abs().max()) == 0.0, f"{what}: {n} has no gradient on one rank but max |g| = {float(a.abs().max()):.3e} on two"
            continue
        s = float(b.abs().max()) + floor
        d = (a - b).abs()
        i = int(d.argmax())
        r = float(d[i]) / s
        if r > worst[0]:
            worst = (r, n, float(a[i]), float(b[i]), s)
    assert worst[0] <= tol, (f"{what}: worst parameter {worst[1]}: 2-rank mean {worst[2]:.9e} vs 1-rank {worst[3]:.9e} "
                             f"(|d| = {worst[0]:.3e} of the parameter's max |g| + 1e-4 max|g_model| = {worst[4]:.3e}; bar {tol:.1e})")
    return worst[0]


@pytest.mark.parametrize("deterministic", [True, False], ids=["fixed_order", "default"])
def test_two_rank_step_equals_single_rank_on_the_whole_batch(tmp_path, deterministic):
    """(1) the REDUCED GRADIENT BUFFER (2-rank SUM x 1/2) equals the whole-batch gradient of one rank, compared before the
    optimiser touches it -- with the fixed-order (ops.DETERMINISTIC) backward and with the default one (split-K atomics,
    deferred unpack tables flushed per bucket from whichever stream completes it: the path a real multi-GPU run takes);
    (2) the two-rank parameters after one fused step equal fp64 AdamW applied to THAT gradient with grad_scale = 1/2 (Adam turns
    the sign of a ~0 gradient into a full +-lr step, so parameters are never compared across summation orders);
    (3) a second pass from the two-rank parameters gives the same agreement (buffers, tables and streams are reused)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, deterministic), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    from adm_amd import ops
    from adm_amd.optim import BucketedGradReducer, FlatParams
    old = ops.DETERMINISTIC
    ops.DETERMINISTIC = bool(deterministic)
    try:
        dev = torch.device("cuda:0")
        dpm = _build(dev)
        flat = FlatParams(dpm)
        red = BucketedGradReducer(flat)
        assert not red.active
        names = [n for n, p in dpm.named_parameters() if p.requires_grad]
        x0, noise, t = _data(dev)
        assert torch.equal(got["start"], flat.flat.detach().cpu()), "the two sides do not start from the same parameters"
        tol = 2e-5 if deterministic else 1e-4
        want1 = _grads(dpm, flat, red, x0, noise, t)
        e1 = _compare_grads(got["g1"], want1, flat, names, 2, "step 1", tol)
        # (2) the optimiser on the two-rank side: fp64 AdamW, step 1 (m = g, v = g^2 after bias correction)
        g = got["g1"].double() * 0.5
        norm = float(g.norm())
        assert abs(got["n1"] - norm) <= 1e-5 * norm, (got["n1"], norm)
        g = g * min(1.0, 1.0 / (norm + 1e-6))
        p0 = got["start"].double()
        want_p = p0 * (1 - LR * 1e-4) - LR * g / (g.abs() + 1e-8)
        dp = (got["p1"].double() - want_p).abs()
        i = int(dp.argmax())
        assert float(dp[i]) <= 2e-6, (f"fused clip + AdamW with grad_scale 1/2: parameter #{i} is {float(got['p1'][i]):.9e}, fp64 AdamW on the "
                                       f"reduced gradient gives {float(want_p[i]):.9e} (gradient {float(g[i]):.3e})")
        assert float((got["p1"] - got["start"]).abs().max()) > 1e-4          # the optimiser did move the parameters
        # (3) second pass from the two-rank parameters
        flat.flat.copy_(got["p1"].to(dev))
        ops.repack_all()
        want2 = _grads(dpm, flat, red, x0, noise, t)
        e2 = _compare_grads(got["g2"], want2, flat, names, 2, "step 2", tol)
        print(f"two-rank vs one-rank reduced gradients: worst |d| / max|g_param| = {e1:.2e} (step 1), {e2:.2e} (step 2); "
              f"table uploads on rank 0: {got['uploads']}")
    finally:
        ops.DETERMINISTIC = old


def _torchrun(script_args, tmp_path, timeout=240):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, ADM_DIST_BACKEND="gloo", ADM_LOCAL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=root)


def test_bench_two_ranks_prints_one_whole_job_line(tmp_path):
    """bench.py under the driver's launch line with N=2 (gloo on one card): rank 0 prints exactly one JSON line whose
    value is the whole-job rate (2 x per-rank batch / max-over-ranks time) and n_gpus = 2."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = _torchrun(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--small", "--batch", "8"], tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    assert abs(out["value"] - 16 / (out["ms_per_step"] / 1e3)) <= 0.01 * out["value"]
    assert out["sample_images_per_sec"] > 0 and "cpu_baseline" not in out
    assert out["roofline"]["launches_per_step"] > 0


def test_trainer_two_ranks_checkpoint(tmp_path):
    """train_uncond_dpm.py with 2 ranks: global batch split, rank-0 checkpoint/EMA, both ranks finish."""
    import yaml
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.load(open(os.path.join(root, "configs/cifar10/ddm_uncond_const_uncond_unet.yaml")), Loader=yaml.SafeLoader)
    cfg["model"]["unet"].update(model_channels=64, num_blocks=1)
    cfg["data"]["batch_size"] = 8
    res = str(tmp_path / "run")
    cfg["trainer"].update(results_folder=res, train_num_steps=2, save_and_sample_every=2, log_freq=1, test_before=False,
                          gradient_accumulate_every=2, ema_update_after_step=0, ema_update_every=1)
    path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    r = _torchrun(["train_uncond_dpm.py", "--cfg", path], tmp_path)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "[Train Step] 2/2" in r.stdout and "training complete" in r.stdout
    ck = torch.load(os.path.join(res, "model-1.pt"), map_location="cpu", weights_only=True)
    assert ck["step"] == 2 and "ema_model.model.model.enc.32x32_conv.weight" in ck["ema"]
    assert os.path.exists(os.path.join(res, "sample-1.png"))


def test_bench_runs_over_a_one_rank_rccl_communicator():
    """The REAL backend on the driver's one-GPU box: `bench.py --small` with ADM_FORCE_DIST=1 builds a 1-rank `nccl`
    (= RCCL) process group, so the hooks, the side stream, the bucketed ncclAllReduce calls and the exposed-wait diagnostic
    all run on hardware; the JSON line must say which backend and how many ranks RCCL saw."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--small", "--steps", "3", "--warmup", "1", "--batch", "16",
                        "--no-cpu-baseline", "--no-sample"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    d = out["dist"]
    assert d["backend"] == "nccl" and d["nranks"] == 1 and d["reducer_active"] is True and d["buckets"] >= 1
    assert d["allreduce_exposed_ms_per_step"] >= 0.0
    assert out["value"] > 0 and out["config"]["valid"] is False       # --small is a debug configuration
    assert 0 < out["roofline"]["frac"] <= 1.0 and 0 < out["roofline"]["step"]["frac"] <= 1.0
