"""Dotted-path alias so YAML `class_name: unet.cond_unet_sd.Unet` resolves to the HIP implementation."""
from adm_amd.unet.cond_unet_sd import *  # noqa: F401,F403
from adm_amd.unet.cond_unet_sd import Unet  # noqa: F401
