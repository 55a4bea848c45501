"""DDM analytic-schedule diffusion wrapper on the HIP hot path.

Mirrors the constructor / ``training_step`` / ``sample`` contract that the reference's drivers call
(SURVEY.md section 8b): ``DDPM(model=unet, cfg=model_cfg, **model_cfg)``,
``training_step(batch) -> (loss, log_dict)``, ``sample(batch_size) -> [B,C,H,W] in [0,1]`` (fp64 for
the deterministic sampler, as the reference returns), buffer ``eps``, attributes ``image_size``,
``channels``.  Skeleton: /root/reference/ddm/ddm_const_2.py:43-389; the 'const' (sqrt t) arithmetic:
/root/reference/ddm/ddm_const.py:284-303, 305-364, 367-476.

Differences, all documented in DESIGN.md:
  * every RNG draw (t, noise, x_T, sampler epsilons) can be injected for parity tests;
  * the LPIPS/VGG16 term needs fetched weights and cannot run offline: ``loss_vlb`` is 0 (the
    reference itself crashes in that configuration, ddm_const_2.py:251);
  * ``use_augment`` runs the AugmentPipe on the GPU (adm_amd/ddm/augment.py) without the reference's per-step host
    read-back of the padding margin; its draws are injectable (``augment_draws=``).
"""
from __future__ import annotations

import warnings
from typing import Optional

import torch
import torch.nn as nn

from .. import ops


def _cfg_get(cfg, key, default):
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class DDPMBase(nn.Module):
    SCHEDULE = "const"          # 'const' -> g(t) = sqrt(t) ; 'const_2' -> g(t) = t
    DEFAULT_EPS = 1e-4
    USES_LPIPS = True           # the pixel-space p_losses has the LPIPS term; LatentDiffusion's does not
    AUGMENT_P = 0.15            # AugmentPipe probability multiplier: 0.15 in ddm_const.py:179, 0.12 in ddm_const_2.py:112
    SUPPORTS_L1 = False         # use_l1 (L1 twins of the SSE terms) exists in the latent p_losses only (ddm_const_2.py:556-559)
    CLIP_IN_SAMPLER = True      # pixel-space 'const' sampler clamps the predicted x0 (ddm_const.py:453-454); latent ones never do

    def __init__(self, model, *, image_size, sampling_timesteps=None, loss_type="l2", objective="pred_noise",
                 beta_schedule="cosine", clip_x_start=True, input_keys=("image",), start_dist="normal",
                 sample_type="naive", perceptual_weight=1.0, use_l1=False, **kwargs):
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", [])
        only_model = kwargs.pop("only_model", False)
        cfg = kwargs.pop("cfg", None)
        super().__init__()
        self.model = model
        self.channels = self.model.channels
        self.self_condition = self.model.self_condition
        self.input_keys = list(input_keys)
        self.cfg = cfg if cfg is not None else {}
        self.scale_input = _cfg_get(cfg, "scale_input", 1)
        self.register_buffer("eps", torch.tensor(float(_cfg_get(cfg, "eps", self.DEFAULT_EPS))))
        self.sigma_min = _cfg_get(cfg, "sigma_min", 1e-2)
        self.sigma_max = _cfg_get(cfg, "sigma_max", 1)
        self.weighting_loss = _cfg_get(cfg, "weighting_loss", False)
        self.clip_x_start = clip_x_start
        self.image_size = image_size
        self.objective = objective
        if start_dist not in ("normal", "uniform"):
            raise AssertionError("start_dist must be 'normal' or 'uniform'")
        self.start_dist = start_dist
        self.loss_type = loss_type
        self.sampling_timesteps = 10 if sampling_timesteps is None else sampling_timesteps
        if use_l1 and not self.SUPPORTS_L1:
            raise NotImplementedError("use_l1 is implemented for the latent wrappers only (no pixel-space DDM config sets it)")
        self.use_l1 = use_l1
        self.perceptual_weight = perceptual_weight
        if perceptual_weight > 0 and self.USES_LPIPS:
            warnings.warn("adm_amd: the LPIPS term (perceptual_weight > 0) needs VGG16 weights that cannot be fetched "
                          "offline; loss_vlb is 0 (see DESIGN.md)", stacklevel=2)
        self.use_augment = bool(_cfg_get(cfg, "use_augment", False))
        if self.use_augment:      # ddm_const.py:179-180 (p = 0.15) / ddm_const_2.py:112-113 (p = 0.12)
            from .augment import AugmentPipe
            self.augment = AugmentPipe(p=self.AUGMENT_P, xflip=1e8, yflip=1, scale=1, rotate_frac=1, aniso=1, translate_frac=1)
        self._eps_f = float(self.eps)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys, only_model)

    # ------------------------------------------------------------------ checkpoints
    def init_from_ckpt(self, path, ignore_keys=(), only_model=False, use_ema=False):
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if "ema" in sd and use_ema:
            sd = {(k[10:] if k.startswith("ema_model.") else k): v for k, v in sd["ema"].items()}
        elif "model" in sd:
            sd = sd["model"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        target = self.model if only_model else self
        missing, unexpected = target.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")

    # ------------------------------------------------------------------ schedule pieces
    @property
    def _sched(self) -> int:
        return 0 if self.SCHEDULE == "const" else 1

    def _g(self, t):
        return torch.sqrt(t) if self.SCHEDULE == "const" else t

    def q_sample(self, x_start, noise, t, C=None):
        """x_t = x0 + C t + g(t) eps with C = -x0 (the only C the wrapper ever uses)."""
        return ops.q_sample(x_start, noise, t.to(torch.float32), self._sched)

    def pred_x0_from_xt(self, xt, noise, C, t):
        """x0 = x_t - C t - g(t) eps (ddm_const.py:290-293 / ddm_const_2.py:179-182).  API-parity helper: the samplers
        do this inside their fused HIP step kernels; this standalone form is [B]-broadcast elementwise math."""
        time = t.reshape(C.shape[0], *((1,) * (C.dim() - 1)))
        return xt - C * time - self._g(time) * noise

    def pred_xtms_from_xt(self, xt, noise, C, t, s, epsilon=None):
        """One stochastic reverse step (ddm_const.py:296-303 / ddm_const_2.py:185-197); `epsilon` injectable."""
        time = t.reshape(C.shape[0], *((1,) * (C.dim() - 1)))
        s = s.reshape(C.shape[0], *((1,) * (C.dim() - 1)))
        if self.SCHEDULE == "const":
            mean = xt + C * (time - s) - C * time - s / torch.sqrt(time) * noise
            sigma = torch.sqrt(s * (time - s) / time)
        else:
            mean = xt - C * s - (2 * s * time - s ** 2) / time * noise
            sigma = torch.sqrt(2 * s * time - s ** 2) * (time - s) / time
        if epsilon is None:
            epsilon = torch.randn_like(mean)
        return mean + sigma * epsilon

    def loss_weights(self, t):
        eps = self._eps_f
        if not self.weighting_loss:
            return torch.ones_like(t), torch.ones_like(t)
        if self.SCHEDULE == "const":       # ddm_const.py:336-338
            return (t ** 2 - t + 1) / t, (t ** 2 - t + 1) / (1 - t + eps)
        return ((t - 1) / t) ** 2 + 1, (t / (1 - t + eps)) ** 2 + 1      # ddm_const_2.py:228-230

    # ------------------------------------------------------------------ training
    def get_input(self, batch):
        for k in self.input_keys:
            if k in batch:
                return batch[k]
        return next(iter(batch.values()))

    def training_step(self, batch, *args, **kwargs):
        z = self.get_input(batch)
        return self(z, *args, **kwargs)

    def forward(self, x, *args, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, **kwargs):
        if self.scale_input != 1:
            x = x * self.scale_input
        if t is None:
            t = torch.rand(x.shape[0], device=x.device) * (1.0 - self._eps_f) + self._eps_f
        return self.p_losses(x, t, *args, noise=noise, **kwargs)

    def p_losses(self, x_start, t, *args, noise: Optional[torch.Tensor] = None, **kwargs):
        if noise is None:
            if self.start_dist == "normal":
                noise = torch.randn_like(x_start)
            else:
                noise = 2 * torch.rand_like(x_start) - 1.0
        if self.use_augment and "augment_labels" not in kwargs:      # ddm_const.py:314-316 / ddm_const_2.py:206-208
            x_start, kwargs["augment_labels"] = self.augment(x_start, draws=kwargs.pop("augment_draws", None))
        kwargs.pop("augment_draws", None)
        x_start = x_start.to(torch.float32).contiguous()
        t = t.to(torch.float32).contiguous()
        x_noisy = self.q_sample(x_start, noise, t)
        C_pred, noise_pred = self.model(x_noisy, t, **kwargs)
        w1, w2 = self.loss_weights(t)
        w = torch.stack([w1, w2], dim=1).contiguous()
        loss_simple_mean, per_sample = ops.ddm_loss(C_pred, noise_pred, x_start, noise, w)
        B, n = x_start.shape[0], x_start[0].numel()
        loss_vlb = torch.zeros((), device=x_start.device)
        loss = loss_simple_mean + loss_vlb
        log = {"train/loss_simple": per_sample.sum() / B / n,
               "train/loss_vlb": loss_vlb / B / n,
               "train/loss": loss.detach() / B / n}
        return loss, log

    # ------------------------------------------------------------------ sampling
    def _use_graph(self) -> bool:
        """HIP-graph replay of the sampler's UNet forward (cfg key ``sample_graph`` or ADM_SAMPLE_GRAPH=1).  The forward
        is ~900 launches; at small batch they are launch-bound and a graph removes the per-launch host cost."""
        import os
        return bool(_cfg_get(self.cfg, "sample_graph", False)) or os.environ.get("ADM_SAMPLE_GRAPH", "0") == "1"

    def _sampling_graph(self, shape, dev):
        """Capture (once per shape and weight version) ``C, noise = model(x_static, t_static)``.  Everything inside is
        device-side and allocation-free at the C-ABI level (torch's capture-time allocations come from the graph's private
        pool), so the capture is legal; the fp64 state update stays outside (its t values are host scalars)."""
        from .. import ops
        cache = self.__dict__.setdefault("_graphs", {})
        key = (shape, str(dev))
        ent = cache.get(key)
        if ent is not None and ent["epoch"] == ops._pack_epoch and not self.training:
            return ent
        x = torch.zeros(shape, dtype=torch.float64, device=dev)
        t = torch.ones((), dtype=torch.float64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):              # warm-up: packs weights, sets kernel attributes, fills allocator pools
            self.model(x, t)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ops.new_amax_pool()                    # the bound vectors of this graph: allocated and zero-filled inside the capture
            C, noise = self.model(x, t)
        ops.new_amax_pool()                        # (eager launches must not write into the graph's private memory)
        ent = {"graph": graph, "x": x, "t": t, "C": C, "noise": noise, "epoch": ops._pack_epoch}
        cache[key] = ent
        return ent

    def _start_noise(self, shape, dev):
        """x_T of the samplers: N(0,1) or U(-1,1) by ``start_dist`` (ddm_const_2.py:305-310, 353-358)."""
        if self.start_dist == "normal":
            return torch.randn(shape, device=dev, dtype=torch.float64)
        return 2 * torch.rand(shape, device=dev, dtype=torch.float64) - 1.0

    def t_steps(self):
        n = self.sampling_timesteps
        i = torch.arange(n, dtype=torch.float64)
        end = self.sigma_min ** 2 if self.SCHEDULE == "const" else 1.0 / n
        ts = self.sigma_max + i / (n - 1) * (end - self.sigma_max)
        return torch.cat([ts, torch.zeros(1, dtype=torch.float64)])

    @torch.no_grad()
    def sample(self, batch_size=16, up_scale=1, cond=None, denoise=True, x_T: Optional[torch.Tensor] = None):
        if cond is not None:
            raise NotImplementedError("conditional sampling is not on the unconditional hot path")
        h, w = self.image_size
        shape = (batch_size, self.channels, h, w)
        sample_type = _cfg_get(self.cfg, "sample_type", "deterministic")
        if sample_type == "deterministic":
            return self.sample_fn_d(shape, x_T=x_T)
        if sample_type == "stochastic":
            return self.sample_fn_s(shape, x_T=x_T)
        raise NotImplementedError(sample_type)

    @torch.no_grad()
    def sample_fn_d(self, shape, up_scale=1, unnormalize=True, cond=None, denoise=False, x_T=None,
                    return_traj=False):
        """10-step deterministic sampler, fp64 state (ddm_const.py:424-476 / ddm_const_2.py:338-389)."""
        dev = self.eps.device
        ts = self.t_steps()
        if x_T is None:
            x_T = self._start_noise(shape, dev)
        x = (x_T.to(device=dev, dtype=torch.float64) * float(ts[0])).contiguous()
        clip = self.clip_x_start and self.SCHEDULE == "const" and self.CLIP_IN_SAMPLER     # const_2 / latent: never clamps x0
        traj = []
        n = len(ts) - 1
        g = self._sampling_graph(tuple(shape), dev) if (self._use_graph() and cond is None) else None
        if g is not None:           # the UNet forward replays from a captured HIP graph on static buffers
            g["x"].copy_(x)
            x = g["x"]
        for i in range(n):
            t_cur, t_next = float(ts[i]), float(ts[i + 1])
            if g is not None:
                g["t"].fill_(t_cur)
                g["graph"].replay()
                C, noise = g["C"], g["noise"]
            elif cond is not None:     # conditional denoiser: model(x, t, cond)  (ddm_const_2.py:712-715)
                C, noise = self.model(x, torch.tensor(t_cur, dtype=torch.float64, device=dev), cond)
            else:
                C, noise = self.model(x, torch.tensor(t_cur, dtype=torch.float64, device=dev))
            last = unnormalize and i == n - 1
            if return_traj and last:   # trajectory holds the pre-normalisation state
                xc = x.clone()
                ops.sampler_step(xc, C, noise, t_cur, t_next, self._sched, clip, float(self.scale_input), False)
                traj.append(xc)
            ops.sampler_step(x, C, noise, t_cur, t_next, self._sched, clip, float(self.scale_input), last)
            if return_traj and not last:
                traj.append(x.clone())
        if g is not None:           # x is the graph's static input buffer: the next sample() of this shape overwrites it
            x = x.clone()
        return (x, traj) if return_traj else x

    @torch.no_grad()
    def sample_fn_s(self, shape, up_scale=1, unnormalize=True, cond=None, denoise=False, x_T=None, epsilons=None):
        """Stochastic sampler (ddm_const.py:380-422 / ddm_const_2.py:288-336) with the per-step update in one HIP
        kernel on an fp64 state.  (The reference promotes to fp64 on its first update anyway -- cur_time is fp64 for
        'const', the state itself for 'const_2' -- so an fp64 state from the start is the same arithmetic.)"""
        dev = self.eps.device
        n = self.sampling_timesteps
        i = torch.arange(n, dtype=torch.float64)
        ts = self.sigma_max ** 2 + i / (n - 1) * (self.sigma_min ** 2 - self.sigma_max ** 2)
        ts = torch.cat([ts, torch.zeros(1, dtype=torch.float64)])
        steps = (-torch.diff(ts)).tolist()
        B = shape[0]
        if x_T is None:
            x_T = self._start_noise(shape, dev)
        img = x_T.to(device=dev, dtype=torch.float64).contiguous()
        if self.SCHEDULE != "const":
            img = img * self.sigma_max
        cur = 1.0
        for k in range(n):
            s = cur if k == n - 1 else steps[k]
            t_vec = torch.full((B,), cur, dtype=torch.float64, device=dev)
            s_vec = torch.full((B,), s, dtype=torch.float64, device=dev)
            C, noise = self.model(img, t_vec)
            z = (epsilons[k].to(device=dev, dtype=torch.float64) if epsilons is not None
                 else torch.randn(shape, device=dev, dtype=torch.float64)).contiguous()
            ops.sampler_step_stochastic(img, C, noise, z, t_vec, s_vec, self._sched, self.clip_x_start,
                                        float(self.scale_input), unnormalize and k == n - 1)
            cur = cur - s
        if not unnormalize:
            img = img.clamp(-self.scale_input, self.scale_input) / self.scale_input
        return img
