"""Dotted-path alias so YAML `class_name: ddm.ddm_const.DDPM` resolves to the HIP implementation."""
from adm_amd.ddm.ddm_const import DDPM, LatentDiffusion  # noqa: F401
