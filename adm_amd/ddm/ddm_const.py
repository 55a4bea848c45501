"""HIP drop-in for /root/reference/ddm/ddm_const.py: x_t = x0 + C t + sqrt(t) eps (eps default 1e-4).

``DDPM``            pixel-space wrapper (arithmetic: ddm_const.py:284-303, 305-364, 367-476).
``LatentDiffusion`` the sqrt(t) schedule in the latent space of a frozen KL autoencoder, optionally CONDITIONAL (the
                    super-resolution recipe, configs/super-resolution/div2k_cond_ddm_const_ldm.yaml:1-21).  The fork rewrote its
                    ddm_const.LatentDiffusion into a pytorch_lightning / nuScenes shell that cannot be imported; what its
                    drivers need is the upstream-shaped class, which is ddm_const_2.LatentDiffusion (ddm_const_2.py:393-737:
                    constructor, std-rescaling, get_input, training_step with `cond`, p_losses with the L1 terms, sample with
                    `cond`) evaluated with this module's schedule.  The fork's own latent samplers (ddm_const.py:830-889) use the
                    same updates: deterministic x += (t' - t)(C + eps / (sqrt t + sqrt t')) == x0 + C t' + sqrt(t') eps, no clamps.
"""
from . import ddm_const_2 as _c2
from .ddpm import DDPMBase


class DDPM(DDPMBase):
    SCHEDULE = "const"
    DEFAULT_EPS = 1e-4


class LatentDiffusion(_c2.LatentDiffusion):
    SCHEDULE = "const"
    DEFAULT_EPS = 1e-4
    AUGMENT_P = 0.15
