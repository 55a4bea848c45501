"""Dotted-path alias so YAML `class_name: ddm.encoder_decoder.AutoencoderKL` resolves to the HIP implementation."""
from adm_amd.ddm.encoder_decoder import AutoencoderKL, DiagonalGaussianDistribution  # noqa: F401
