"""HIP drop-in for /root/reference/unet/uncond_unet.py: two decoders, 'const' preconditioning (:621-626)."""
from .dhariwal import (Conv2d, DhariwalUNet, GroupNorm, Linear, PositionalEmbedding, SpatialAtt, UNetBlock)  # noqa: F401
from .dhariwal import EDMPrecond as _EDMPrecond


class EDMPrecond(_EDMPrecond):
    VARIANT = "uncond_unet"
