"""Dotted-path alias so YAML `class_name: unet.uncond_unet_sd.EDMPrecond` resolves to the HIP implementation."""
from adm_amd.unet.uncond_unet_sd import *  # noqa: F401,F403
from adm_amd.unet.uncond_unet_sd import EDMPrecond  # noqa: F401
