"""HIP drop-in for /root/reference/unet/cond_unet_sd.py: the conditional SR denoiser with ONE decoder; the noise branch is
derived analytically, x2 = (x - (t - 1) x1) / sqrt(t) (:878-882)."""
from .cond_unet import (Attention, BasicAttetnionLayer, Block, Downsample, GaussianFourierProjection, LayerNorm,  # noqa: F401
                        LinearAttention, Mlp, PreNorm, RelationNet, Residual, ResnetBlock, SpatialAtt, Upsample,
                        WeightStandardizedConv2d)
from .cond_unet import Unet as _Unet


class Unet(_Unet):
    TWO_DECODERS = False
