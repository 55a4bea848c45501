"""Oracle (CPU, plain PyTorch) restatement of the reference's analytic-schedule diffusion wrapper.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Two schedules, selected by ``schedule``:
  'const'   x_t = x0 + C t + sqrt(t) eps     /root/reference/ddm/ddm_const.py:284-303, 305-364, 367-476
  'const_2' x_t = x0 + C t + t eps           /root/reference/ddm/ddm_const_2.py:173-197, 199-258, 275-389
with C = -x0.  The perceptual (LPIPS/VGG16) term of the reference loss needs fetched weights and
has no oracle: everything here is the ``loss_vlb == 0`` path.

Every function takes ``model_fn(x, t, **kw) -> (C_pred, noise_pred)`` so the same code can be
driven by the oracle UNet, by the imported reference UNet (tools/make_golden.py) or by the HIP UNet.
RNG draws (t, noise, x_T, sampler epsilons) are always injectable.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

Tensor = torch.Tensor


def _bc(t: Tensor, ref: Tensor) -> Tensor:
    return t.reshape(ref.shape[0], *((1,) * (ref.dim() - 1)))


def _g(schedule: str, t: Tensor) -> Tensor:
    """noise gain g(t): sqrt(t) for 'const', t for 'const_2'."""
    return torch.sqrt(t) if schedule == "const" else t


def q_sample(schedule: str, x0: Tensor, noise: Tensor, t: Tensor, C: Tensor) -> Tensor:
    """ddm_const.py:284-287 / ddm_const_2.py:173-176."""
    time = _bc(t, C)
    return x0 + C * time + _g(schedule, time) * noise


def pred_x0_from_xt(schedule: str, xt: Tensor, noise: Tensor, C: Tensor, t: Tensor) -> Tensor:
    """ddm_const.py:290-293 / ddm_const_2.py:179-182."""
    time = _bc(t, C)
    return xt - C * time - _g(schedule, time) * noise


def pred_xtms_from_xt(schedule: str, xt, noise, C, t, s, epsilon) -> Tensor:
    """Stochastic reverse step with the N(0,1) draw ``epsilon`` injected.
    ddm_const.py:296-303 / ddm_const_2.py:185-197."""
    time, s = _bc(t, C), _bc(s, C)
    if schedule == "const":
        mean = xt + C * (time - s) - C * time - s / torch.sqrt(time) * noise
        sigma = torch.sqrt(s * (time - s) / time)
    else:
        mean = xt - C * s - (2 * s * time - s ** 2) / time * noise
        sigma = torch.sqrt(2 * s * time - s ** 2) * (time - s) / time
    return mean + sigma * epsilon


def loss_weights(schedule: str, t: Tensor, eps: float, weighting_loss: bool = True):
    """ddm_const.py:335-341 / ddm_const_2.py:227-233."""
    if not weighting_loss:
        one = torch.ones_like(t)
        return one, one
    if schedule == "const":
        return (t ** 2 - t + 1) / t, (t ** 2 - t + 1) / (1 - t + eps)
    return ((t - 1) / t) ** 2 + 1, (t / (1 - t + eps)) ** 2 + 1


def p_losses(schedule: str, model_fn: Callable, x0: Tensor, t: Tensor, noise: Tensor, eps: float,
             weighting_loss: bool = True, **model_kw):
    """ddm_const.py:305-364 / ddm_const_2.py:199-258 with perceptual_weight = 0, use_l1 = False,
    loss_main = ddm.loss.MSE_Loss(reduction='sum') (ddm/loss.py:292-333).
    Returns (loss, log_dict, (x_noisy, C_pred, noise_pred))."""
    C = -1 * x0
    x_noisy = q_sample(schedule, x0, noise, t, C)
    C_pred, noise_pred = model_fn(x_noisy, t, **model_kw)
    w1, w2 = loss_weights(schedule, t, eps, weighting_loss)
    sse = lambda a, b: ((a - b) ** 2).sum(dim=[1, 2, 3])
    loss_simple = w1 * sse(C_pred, C) + w2 * sse(noise_pred, noise)
    B, n = C.shape[0], C[0].numel()
    loss_vlb = torch.zeros((), dtype=loss_simple.dtype)
    loss = loss_simple.sum() / B + loss_vlb
    log = {"train/loss_simple": loss_simple.detach().sum() / B / n,
           "train/loss_vlb": loss_vlb.detach() / B / n,
           "train/loss": loss.detach() / B / n}          # the reference divides by B twice here
    return loss, log, (x_noisy, C_pred, noise_pred)


def draw_t(u: Tensor, eps: float) -> Tensor:
    """DDPM.forward: t = U(0,1) * (1 - eps) + eps  (ddm_const.py:278-279)."""
    return u * (1.0 - eps) + eps


def t_steps_deterministic(schedule: str, n: int, sigma_min: float, sigma_max: float) -> Tensor:
    """fp64 time grid of sample_fn_d incl. the trailing 0.
    ddm_const.py:428-436 (end point sigma_min**2) / ddm_const_2.py:341-349 (end point 1/n)."""
    i = torch.arange(n, dtype=torch.float64)
    end = sigma_min ** 2 if schedule == "const" else 1.0 / n
    ts = sigma_max + i / (n - 1) * (end - sigma_max)
    return torch.cat([ts, torch.zeros(1, dtype=torch.float64)])


def sample_fn_d(schedule: str, model_fn: Callable, x_T: Tensor, n: int, sigma_min: float, sigma_max: float,
                scale_input: float = 1.0, clip_x_start: bool = True, return_traj: bool = False):
    """Deterministic sampler.  ``x_T`` is the unit normal draw (fp64, [B,C,H,W]); the reference
    multiplies it by t_steps[0].  ddm_const.py:424-476 / ddm_const_2.py:338-389.
    Returns fp64 images in [0,1] (and the list of x_next per step when return_traj)."""
    ts = t_steps_deterministic(schedule, n, sigma_min, sigma_max)
    x = x_T.to(torch.float64) * ts[0]
    traj: List[Tensor] = []
    for t_cur, t_next in zip(ts[:-1], ts[1:]):
        C, eps = model_fn(x, t_cur)
        C, eps = C.to(torch.float64), eps.to(torch.float64)
        x0 = x - C * t_cur - eps * _g(schedule, t_cur)
        if schedule == "const" and clip_x_start:      # const_2's sample_fn_d has no clamp
            x0 = x0.clamp(-scale_input, scale_input)
        x = x0 + C * t_next + eps * _g(schedule, t_next)
        traj.append(x)
    img = x.clamp(-scale_input, scale_input) / scale_input
    img = (img + 1) * 0.5
    return (img, traj) if return_traj else img


def sample_fn_s(schedule: str, model_fn: Callable, x_T: Tensor, epsilons: Sequence[Tensor], n: int,
                sigma_min: float, sigma_max: float, scale_input: float = 1.0, clip_x_start: bool = True):
    """Stochastic sampler with the per-step N(0,1) draws injected.
    ddm_const.py:380-422 / ddm_const_2.py:288-336."""
    i = torch.arange(n, dtype=torch.float64)
    ts = sigma_max ** 2 + i / (n - 1) * (sigma_min ** 2 - sigma_max ** 2)
    ts = torch.cat([ts, torch.zeros(1, dtype=torch.float64)])
    steps = -torch.diff(ts)
    B = x_T.shape[0]
    if schedule == "const":
        img = x_T.to(torch.float32)
        cur = torch.ones(B, dtype=torch.float64)
    else:
        img = x_T.to(torch.float64) * sigma_max
        cur = torch.ones(B, dtype=torch.float32)
    for k, step in enumerate(steps):
        s = torch.full((B,), float(step), dtype=torch.float32)
        if k == len(steps) - 1:
            s = cur
        C, noise = model_fn(img, cur)
        x0 = pred_x0_from_xt(schedule, img, noise, C, cur)
        if clip_x_start:
            x0 = x0.clamp(-scale_input, scale_input)
        C = -1 * x0
        img = pred_xtms_from_xt(schedule, img, noise, C, cur, s, epsilons[k].to(torch.float64))
        cur = cur - s
    img = img.clamp(-scale_input, scale_input) / scale_input
    return (img + 1) * 0.5


# ----------------------------------------------------------------------------------------------
# Driver-side schedules that the reference's Trainer uses (train_uncond_dpm.py:169-177, ddm/ema.py)
# ----------------------------------------------------------------------------------------------

def lr_lambda(it: int, lr: float, min_lr: float, train_num_steps: int, warmup: int = 5000) -> float:
    """train_uncond_dpm.py:169-177 WarmUpLrScheduler: linear warm-up then (1-x)^0.96, floored."""
    if it <= warmup:
        return (it + 1) / warmup
    return max((1 - (it - warmup) / train_num_steps) ** 0.96, min_lr / lr)


def ema_decay(step: int, beta: float = 0.9996, update_after_step: int = 10000, inv_gamma: float = 1.0,
              power: float = 2 / 3, min_value: float = 0.0) -> float:
    """ddm/ema.py:141-152 get_current_decay."""
    epoch = max(step - update_after_step - 1, 0.0)
    if epoch <= 0:
        return 0.0
    value = 1 - (1 + epoch / inv_gamma) ** (-power)
    return min(max(value, min_value), beta)


class OracleDDPM:
    """Minimal stateful wrapper used by tests and the cpu_baseline timing leg: holds an oracle UNet
    state dict and exposes training_step / sample with injectable randomness."""

    def __init__(self, unet_sd: Dict[str, Tensor], unet_cfg: dict, schedule: str = "const", *,
                 image_size=(32, 32), sampling_timesteps: int = 10, eps: float = 1e-4, sigma_min: float = 0.01,
                 sigma_max: float = 1.0, weighting_loss: bool = True, scale_input: float = 1.0,
                 clip_x_start: bool = True):
        from . import unet_ref
        self.sd, self.cfg, self.schedule = unet_sd, unet_cfg, schedule
        self.image_size, self.n = tuple(image_size), sampling_timesteps
        self.eps, self.sigma_min, self.sigma_max = eps, sigma_min, sigma_max
        self.weighting_loss, self.scale_input, self.clip_x_start = weighting_loss, scale_input, clip_x_start
        self.training = False
        self._unet = unet_ref

    def model_fn(self, x, t, **kw):
        return self._unet.edm_precond(self.sd, self.cfg, x, t, training=self.training, **kw)

    def training_step(self, batch, t: Optional[Tensor] = None, noise: Optional[Tensor] = None, **kw):
        x0 = batch["image"] * self.scale_input if self.scale_input != 1 else batch["image"]
        if t is None:
            t = draw_t(torch.rand(x0.shape[0]), self.eps)
        if noise is None:
            noise = torch.randn_like(x0)
        loss, log, _ = p_losses(self.schedule, self.model_fn, x0, t, noise, self.eps, self.weighting_loss, **kw)
        return loss, log

    @torch.no_grad()
    def sample(self, batch_size=16, x_T: Optional[Tensor] = None):
        if x_T is None:
            x_T = torch.randn(batch_size, self.cfg["img_channels"], *self.image_size, dtype=torch.float64)
        return sample_fn_d(self.schedule, self.model_fn, x_T, self.n, self.sigma_min, self.sigma_max,
                           self.scale_input, self.clip_x_start)
