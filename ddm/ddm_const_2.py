"""Dotted-path alias so YAML `class_name: ddm.ddm_const_2.DDPM` / `.LatentDiffusion` resolve to the HIP implementation."""
from adm_amd.ddm.ddm_const_2 import DDPM, LatentDiffusion  # noqa: F401
