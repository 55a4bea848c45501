"""HIP drop-in for /root/reference/unet/uncond_unet_sd_3.py: single decoder + skip-tuning ratios (:547-555)."""
from .dhariwal import (Conv2d, DhariwalUNet, GroupNorm, Linear, PositionalEmbedding, SpatialAtt, UNetBlock)  # noqa: F401
from .dhariwal import EDMPrecond as _EDMPrecond


class EDMPrecond(_EDMPrecond):
    VARIANT = "uncond_unet_sd_3"
