"""Dotted-path alias so YAML `class_name: unet.uncond_unet_sd_3.EDMPrecond` resolves to the HIP implementation."""
from adm_amd.unet.uncond_unet_sd_3 import *  # noqa: F401,F403
from adm_amd.unet.uncond_unet_sd_3 import EDMPrecond  # noqa: F401
