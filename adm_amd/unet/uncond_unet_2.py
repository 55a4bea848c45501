"""HIP drop-in for /root/reference/unet/uncond_unet_2.py: two decoders, 'const_2' preconditioning (:623-627)."""
from .dhariwal import (Conv2d, DhariwalUNet, GroupNorm, Linear, PositionalEmbedding, SpatialAtt, UNetBlock)  # noqa: F401
from .dhariwal import EDMPrecond as _EDMPrecond


class EDMPrecond(_EDMPrecond):
    VARIANT = "uncond_unet_2"
