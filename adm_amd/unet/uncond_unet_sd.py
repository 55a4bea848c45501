"""HIP drop-in for /root/reference/unet/uncond_unet_sd.py: single decoder, D_y = (x-(s-1)D_x)/sqrt(s) (:602)."""
from .dhariwal import (Conv2d, DhariwalUNet, GroupNorm, Linear, PositionalEmbedding, SpatialAtt, UNetBlock)  # noqa: F401
from .dhariwal import EDMPrecond as _EDMPrecond


class EDMPrecond(_EDMPrecond):
    VARIANT = "uncond_unet_sd"
