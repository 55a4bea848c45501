"""HIP drop-in for /root/reference/ddm/ddm_const_2.py: x_t = x0 + C t + t eps.

``DDPM``            pixel-space wrapper (ddm_const_2.py:43-389)
``LatentDiffusion`` the same schedule in the latent space of a frozen KL autoencoder (ddm_const_2.py:393-737):
                    encode (no grad) -> [std-rescale] -> p_losses with an extra L1 reconstruction term -> UNet;
                    sample = latent sampler -> un-scale -> decode -> [0, 1].
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .. import ops
from .ddpm import DDPMBase, _cfg_get


class DDPM(DDPMBase):
    SCHEDULE = "const_2"
    DEFAULT_EPS = 1e-4
    AUGMENT_P = 0.12


class LatentDiffusion(DDPM):
    USES_LPIPS = False
    SUPPORTS_L1 = True
    CLIP_IN_SAMPLER = False

    def __init__(self, auto_encoder, scale_factor=1.0, scale_by_std=True, scale_by_softsign=False, input_keys=("image",),
                 sample_type="naive", default_scale=False, *args, **kwargs):
        self.scale_by_std = scale_by_std
        self.scale_by_softsign = scale_by_softsign
        self.default_scale = default_scale
        ckpt_path = kwargs.pop("ckpt_path", None)
        ignore_keys = kwargs.pop("ignore_keys", [])
        only_model = kwargs.pop("only_model", False)
        super().__init__(*args, **kwargs)
        if not scale_by_std:
            self.scale_factor = scale_factor
        else:
            self.register_buffer("scale_factor", torch.tensor(scale_factor))
        if self.scale_by_softsign:
            self.scale_by_std = False
        assert (self.scale_by_std and self.scale_by_softsign) is False
        self.init_first_stage(auto_encoder)
        self.input_keys = list(input_keys)
        self.clip_denoised = False
        assert sample_type in ["naive", "ddim", "dpm"]
        if _cfg_get(self.cfg, "use_disloss", False):
            raise NotImplementedError("use_disloss (image-space distillation through the decoder) is off in every DDM "
                                      "config and is not implemented")
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys, only_model)

    def init_first_stage(self, first_stage_model):
        self.first_stage_model = first_stage_model.eval()
        for param in self.first_stage_model.parameters():
            param.requires_grad = False

    def get_first_stage_encoding(self, encoder_posterior, eps: Optional[torch.Tensor] = None):
        if hasattr(encoder_posterior, "sample"):
            z = encoder_posterior.sample(eps) if eps is not None else encoder_posterior.sample()
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return z.detach()

    @torch.no_grad()
    def on_train_batch_start(self, batch, eps: Optional[torch.Tensor] = None):
        """First batch only: scale_factor = 1 / std of its encodings (ddm_const_2.py:474-493)."""
        if self.scale_by_std and (not self.scale_by_softsign):
            if not self.default_scale:
                assert float(self.scale_factor) == 1., "rather not use custom rescaling and std-rescaling simultaneously"
                x = next(iter(batch.values()))
                z = self.get_first_stage_encoding(self.first_stage_model.encode(x), eps)
                del self.scale_factor
                self.register_buffer("scale_factor", 1. / z.flatten().std())
                print(f"### USING STD-RESCALING ###\nsetting self.scale_factor to {self.scale_factor}")
            else:
                print(f"### USING DEFAULT SCALE {self.scale_factor}")
        else:
            print("### USING SOFTSIGN SCALE !")

    @torch.no_grad()
    def get_input(self, batch, return_first_stage_outputs=False, return_original_cond=False, eps=None):
        assert "image" in self.input_keys
        x = batch["image"]
        cond = batch["cond"] if "cond" in batch else None
        z = self.get_first_stage_encoding(self.first_stage_model.encode(x), eps)
        out = [z, cond, x]
        if return_first_stage_outputs:
            out.extend([x, self.first_stage_model.decode(z)])
        if return_original_cond:
            out.append(cond)
        return out

    def training_step(self, batch, *, eps: Optional[torch.Tensor] = None, **kwargs):
        """`eps` (posterior draw), and via kwargs `t` / `noise`, are injectable for parity tests."""
        z, c, x, *_ = self.get_input(batch, eps=eps)
        if self.scale_by_softsign:
            z = F.softsign(z)
        elif self.scale_by_std:
            z = self.scale_factor * z
        if c is not None:          # conditional denoiser (super-resolution): model(x_t, t, cond)   (ddm_const_2.py:512-524)
            return self(z, c, **kwargs)
        return self(z, **kwargs)

    def p_losses(self, x_start, t, *args, noise: Optional[torch.Tensor] = None, **kwargs):
        """ddm_const_2.py:527-596 (use_l1 = False, use_disloss = False).  The reference multiplies the [B] vector of
        per-sample L1 sums with a [B, 1] rec_weight (:566-568), i.e. a [B, B] outer product whose sum is
        (sum_b L1_b) * (sum_b' -log t_b' / 2); that quirk is kept: every sample's L1 gradient is weighted by the batch-sum
        W of the rec weights."""
        if noise is None:
            noise = torch.randn_like(x_start) if self.start_dist == "normal" else 2 * torch.rand_like(x_start) - 1.0
        x_start = x_start.to(torch.float32).contiguous()
        t = t.to(torch.float32).contiguous()
        x_noisy = self.q_sample(x_start, noise, t)
        C_pred, noise_pred = self.model(x_noisy, t, *args, **kwargs)
        w1, w2 = self.loss_weights(t)
        W = (-torch.log(t) / 2).sum()
        w = torch.stack([w1, w2, W.expand_as(w1)], dim=1).contiguous()
        loss, per_simple, per_l1 = ops.ddm_loss_latent(C_pred, noise_pred, x_start, noise, x_noisy, t, w, self._sched,
                                                       self.use_l1)
        B, n = x_start.shape[0], x_start[0].numel()
        loss_vlb = per_l1.sum() * W
        log = {"train/loss_simple": per_simple.sum() / B / n,
               "train/loss_vlb": loss_vlb / B / n,
               "train/loss": loss.detach() / B / n}
        return loss, log

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample(self, batch_size=16, up_scale=1, cond=None, mask=None, denoise=True, x_T=None, epsilons=None, latent_hw=None):
        if mask is not None:
            raise NotImplementedError("masked (in-painting) sampling is not implemented")
        if cond is not None:       # ddm_const_2.py:599-601: the condition sets the batch size
            batch_size = cond[0].shape[0] if isinstance(cond, (list, tuple)) else cond.shape[0]
        down = self.first_stage_model.down_ratio
        shape = (batch_size, self.channels, self.image_size[0] // down, self.image_size[1] // down)
        if latent_hw is not None:      # sliding-window SR: a border window smaller than the configured image (sample_cond_ldm.py)
            shape = (batch_size, self.channels, int(latent_hw[0]), int(latent_hw[1]))
        sample_type = _cfg_get(self.cfg, "sample_type", "deterministic")
        if sample_type == "deterministic":
            z = self.sample_fn_d(shape, unnormalize=False, x_T=x_T, cond=cond)
        elif sample_type == "stochastic":
            z = self.sample_fn_s(shape, unnormalize=False, denoise=denoise, x_T=x_T, epsilons=epsilons, cond=cond)
        else:
            raise NotImplementedError(sample_type)
        if self.scale_by_std:
            z = 1. / self.scale_factor * z
        elif self.scale_by_softsign:
            z = z / (1 - z.abs())
        x_rec = self.first_stage_model.decode(z.to(torch.float32))
        return torch.clamp((x_rec + 1) * 0.5, min=0., max=1.)

    @torch.no_grad()
    def sample_fn_s(self, shape, up_scale=1, unnormalize=True, cond=None, denoise=False, x_T=None, epsilons=None):
        """Latent stochastic sampler (ddm_const_2.py:624-675): uniform steps 1/N, with `denoise` the last one split into
        (1/N - eps, eps); C re-derived from the predicted x0 (no clamp).  Per-step update = the fused fp64 HIP kernel
        (the reference keeps an fp32 state here; fp64 only lowers the rounding error)."""
        dev = self.eps.device
        n = self.sampling_timesteps
        steps = [1.0 / n] * n
        if denoise:
            e = float(self.eps)
            steps = steps[:-1] + [steps[-1] - e, e]
        B = shape[0]
        if x_T is None:
            x_T = torch.randn(shape, device=dev) if self.start_dist == "normal" else 2 * torch.rand(shape, device=dev) - 1.
        img = x_T.to(device=dev, dtype=torch.float64).contiguous()
        cur = 1.0
        for k, s in enumerate(steps):
            if k == len(steps) - 1:
                s = cur
            t_vec = torch.full((B,), cur, dtype=torch.float64, device=dev)
            s_vec = torch.full((B,), s, dtype=torch.float64, device=dev)
            C, noise = self.model(img, t_vec, cond) if cond is not None else self.model(img, t_vec)
            z = (epsilons[k].to(device=dev, dtype=torch.float64) if epsilons is not None
                 else torch.randn(shape, device=dev, dtype=torch.float64)).contiguous()
            # scale_by_softsign: the predicted x0 is clamped to +-0.987654321 before C is re-derived (ddm_const_2.py:661-665)
            ops.sampler_step_stochastic(img, C, noise, z, t_vec, s_vec, self._sched, bool(self.scale_by_softsign),
                                        0.987654321 if self.scale_by_softsign else 1.0, False)
            cur = cur - s
        if self.scale_by_softsign:
            img.clamp_(-0.987654321, 0.987654321)
        if unnormalize:
            img = (img + 1) * 0.5
        return img
