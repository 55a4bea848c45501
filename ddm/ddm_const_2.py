"""Dotted-path alias so YAML `class_name: ddm.ddm_const_2.DDPM` resolves to the HIP implementation."""
from adm_amd.ddm.ddm_const_2 import DDPM  # noqa: F401
