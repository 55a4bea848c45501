"""Optimiser-side hot path: flat parameter/gradient buffers, bucketed RCCL gradient all-reduce
overlapped with backward, and one fused HIP pass for clip + AdamW + EMA.

Replaces, for the reference's Trainer loop (/root/reference/train_uncond_dpm.py:280-310):
``accelerator.clip_grad_norm_(params, 1.0)``, ``opt.step()`` (torch.optim.AdamW, lr, wd=1e-4),
``ema.update()`` (/root/reference/ddm/ema.py:158-188) and the DDP gradient mean that
``accelerator.prepare(model)`` was meant to provide (the reference bypasses DDP.forward, SURVEY.md
section 5.8; this implements the intended synchronous mean).

Layout in HBM: ONE contiguous fp32 buffer per role (params, grads, exp_avg, exp_avg_sq, ema); every
nn.Parameter is a view into the params buffer and its ``.grad`` a view into the grads buffer, in
registration order.  Buckets are contiguous slices of the grads buffer, so a bucket all-reduce needs
no packing, and the optimiser is a single launch over 216 M elements.
"""
from __future__ import annotations

import time
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import hip, ops
from .hip import call, ptr


class FlatParams:
    """Re-homes all parameters of ``module`` into one flat buffer (and grads into another)."""

    def __init__(self, module: nn.Module, align: int = 64):
        self.params: List[nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.dtype == torch.float32 for p in self.params), "fp32 master parameters expected"
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=dt)
        self.grad = torch.zeros(off, device=dev, dtype=dt)
        for p, o in zip(self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view(p.shape)
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
            p._adm_direct = True        # ops.* backward kernels may accumulate straight into p.grad

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):       # keep .grad pointing into the flat buffer
            p._adm_uses = 0                               # (ops._mark_uses / ops._notify: graph nodes still to run per parameter)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class BucketedGradReducer:
    """Gradient mean across ranks, bucketed and overlapped with backward.

    Buckets are contiguous slices of ``flat.grad`` built in REVERSE registration order (gradients
    become ready roughly output-to-input: dec2 -> dec -> decouple -> enc -> map), sized for xGMI's
    point-to-point links rather than NVSwitch (default 64 MiB: ~14 buckets for the 864 MB model).
    When the last gradient of a bucket has been accumulated, the bucket's all-reduce (RCCL SUM; the
    1/world factor is folded into the optimiser's grad_scale) is enqueued on a side stream that
    waits on an event recorded on the compute stream.  ``finish()`` makes the compute stream wait
    for the side stream.  Works on CPU tensors with gloo too (synchronously), which is how the
    world_size-2 tests cover it."""

    def __init__(self, flat: FlatParams, bucket_bytes: int = 64 << 20, group=None, force: bool = False):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # ``force`` runs the full hook / side-stream / all-reduce machinery even at world size 1
        # (a 1-rank RCCL communicator): lets a single-GPU box exercise the multi-GPU code path.
        self.active = dist.is_initialized() and (self.world > 1 or force)
        self.cuda = flat.grad.is_cuda
        self.side = torch.cuda.Stream() if (self.cuda and self.active) else None
        # streams on which gradients of the running pass were announced (backward nodes run on the stream of their forward op:
        # the caller's, the second decoder's, the weight-gradient side stream); a bucket's all-reduce is ordered after all of
        # them.  Collected per pass -- NOT captured at construction: backward() may be called from any stream
        self.streams = {}
        self.buckets = []             # (start, end, n_params)
        self.param_bucket = {}
        cap = max(1, bucket_bytes // 4)
        end = flat.numel
        cur_lo, count, members = end, 0, []
        for idx in range(len(flat.params) - 1, -1, -1):
            lo = flat.offsets[idx]
            members.append(idx)
            count += 1
            cur_lo = lo
            if end - cur_lo >= cap or idx == 0:
                b = len(self.buckets)
                self.buckets.append((cur_lo, end, count))
                for m in members:
                    self.param_bucket[m] = b
                end, count, members = cur_lo, 0, []
        self.pending = [0] * len(self.buckets)
        self.seen = [False] * len(flat.params)
        self.enabled = True
        self.handles = []
        self.exposed_ms, self.finishes = 0.0, 0     # diagnostics: time finish() waited for the all-reduces (bench.py "dist")
        self.measure = False
        if self.active:
            for idx, p in enumerate(flat.params):
                hook = self._make_hook(idx)
                p.register_post_accumulate_grad_hook(hook)     # gradients that arrive through autograd
                p._adm_grad_sink = hook                         # gradients the HIP kernels accumulate directly

    def _make_hook(self, idx):
        def hook(_p):
            if not self.enabled or self.seen[idx]:
                return
            # once per parameter and step: a parameter whose gradient a HIP kernel accumulated directly is announced
            # through _adm_grad_sink, and autograd may STILL fire its post-accumulate hook for the (undefined) gradient
            # the backward returned -- counting both launched every bucket twice (harmless at world size 1, a doubled
            # gradient sum at world size > 1)
            self.seen[idx] = True
            if self.cuda:
                st = torch.cuda.current_stream()
                self.streams[st.cuda_stream] = st
            b = self.param_bucket[idx]
            self.pending[b] += 1
            if self.pending[b] == self.buckets[b][2]:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        view = self.flat.grad[lo:hi]
        # weight / GroupNorm-parameter gradients of the layers seen so far may still sit in their workspaces (ops: deferred
        # unpack): scatter them now, one launch for everything since the previous bucket
        ops.flush_deferred_unpack()
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                # the hook may fire on the weight-gradient side stream (ops.SIDE_WGRAD) while other gradients of this
                # bucket were written on the main stream, or the other way round: order the all-reduce after BOTH
                if ops._side_stream is not None:
                    self.side.wait_stream(ops._side_stream)
                if ops._branch_stream is not None:
                    self.side.wait_stream(ops._branch_stream)
                for st in self.streams.values():
                    self.side.wait_stream(st)
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def finish(self):
        """Call after backward: launches any bucket whose hooks did not all fire (parameters without a
        gradient this step) and joins the side stream."""
        ops.join_side_streams()
        if self.active and self.enabled:
            for b in range(len(self.buckets)):
                if self.pending[b] != self.buckets[b][2]:
                    self._launch(b)
            t0 = None
            if self.measure and self.cuda:
                # exposed communication: drain the compute stream first, then time how long the collectives still need
                torch.cuda.current_stream().synchronize()
                t0 = time.perf_counter()
            for h in self.handles:
                h.wait()
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
            if t0 is not None:
                torch.cuda.current_stream().synchronize()
                self.exposed_ms += (time.perf_counter() - t0) * 1e3
                self.finishes += 1
        self.handles = []
        self.streams = {}
        self.pending = [0] * len(self.buckets)
        self.seen = [False] * len(self.flat.params)


class FusedAdamWEMA:
    """clip_grad_norm_(max_norm) + AdamW(lr, betas, eps, weight_decay) + EMA lerp in two HIP launches
    over the flat buffers (sum of squares, then the update)."""

    def __init__(self, flat: FlatParams, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, max_norm=1.0,
                 ema: bool = False):
        self.flat, self.lr, self.betas, self.eps, self.wd, self.max_norm = flat, lr, betas, eps, weight_decay, max_norm
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.ema = flat.flat.clone() if ema else None
        self.sumsq = torch.zeros(1, device=flat.flat.device, dtype=torch.float64)
        # per-workgroup partials of the gradient norm, combined in a fixed order: the clip factor is bitwise reproducible
        self.partials = torch.zeros(hip.lib().adm_sumsq_blocks(flat.numel), device=flat.flat.device, dtype=torch.float64)
        self.step_count = 0

    def step(self, lr: Optional[float] = None, grad_scale: float = 1.0, ema_decay: Optional[float] = None):
        """grad_scale multiplies the raw gradient buffer (1/world after a SUM all-reduce, 1/accum for
        gradient accumulation).  ema_decay None = leave the EMA untouched this step; 0 = copy."""
        f = self.flat
        ops.join_side_streams()        # weight gradients are produced on a side stream (ops.SIDE_WGRAD)
        self.step_count += 1
        self.sumsq.zero_()
        call("adm_sumsq", ptr(f.grad), ptr(self.sumsq), ptr(self.partials), f.numel)
        ema_ptr = ptr(self.ema) if (self.ema is not None and ema_decay is not None) else None
        call("adm_adamw_step", ptr(f.flat), ptr(f.grad), ptr(self.m), ptr(self.v), ema_ptr, ptr(self.sumsq), f.numel,
             float(self.lr if lr is None else lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
             float(self.wd), float(self.max_norm), self.step_count, float(ema_decay if ema_decay is not None else 0.0),
             float(grad_scale))
        ops.repack_all()             # parameters changed through raw pointers: refresh every packed operand (1 launch)

    def grad_norm(self, grad_scale: float = 1.0) -> float:
        return float(self.sumsq.sqrt()) * grad_scale


def lr_lambda(it: int, lr: float, min_lr: float, train_num_steps: int, warmup: int = 5000) -> float:
    """WarmUpLrScheduler of the reference's Trainer (train_uncond_dpm.py:169-177)."""
    if it <= warmup:
        return (it + 1) / warmup
    return max((1 - (it - warmup) / train_num_steps) ** 0.96, min_lr / lr)


def ema_decay_at(step: int, beta: float = 0.9996, update_after_step: int = 10000, inv_gamma: float = 1.0,
                 power: float = 2 / 3, min_value: float = 0.0) -> float:
    """EMA.get_current_decay (ddm/ema.py:141-152)."""
    epoch = max(step - update_after_step - 1, 0.0)
    if epoch <= 0:
        return 0.0
    return min(max(1 - (1 + epoch / inv_gamma) ** (-power), min_value), beta)
