"""Closed-form deterministic parameter / input fill (TEST INFRASTRUCTURE ONLY).

The reference zero-initialises conv1 / proj / map_augment (unet/uncond_unet.py:471-477), so a
default-initialised model never exercises them.  Every parity test therefore overwrites every
parameter with the integer-hash fill below, which is reproducible bit-for-bit on any machine
(pure uint64 arithmetic, no libm), so golden fixtures only need to store expected OUTPUTS.
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np
import torch


def _hash_uniform(n: int, seed: int) -> np.ndarray:
    """n floats in (-1, 1), uniform-ish, from a 32-bit integer mix of (index, seed)."""
    h = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(seed & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x45D9F3B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return (h.astype(np.float64) + 0.5) / 2147483648.0 - 1.0


def hash_tensor(shape, tag: str, scale: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    """Deterministic tensor of ``shape``; ``tag`` selects the stream."""
    n = int(np.prod(shape)) if len(shape) else 1
    v = _hash_uniform(n, zlib.crc32(tag.encode())) * scale
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def fill_value(name: str, shape: Tuple[int, ...]) -> torch.Tensor:
    """Value for state_dict entry ``name`` (reference naming)."""
    if name.endswith("resample_filter"):
        return torch.full(shape, 0.25)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = ".norm" in name or "out_norm" in name
    if leaf == "weight" and is_norm:
        return 1.0 + hash_tensor(shape, name, 0.2)
    if leaf == "bias":
        return hash_tensor(shape, name, 0.1)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        s = (1.0 / fan_in) ** 0.5          # var = 1/(3 fan_in): the reference's Dhariwal init scale
        if ".q_conv" in name or ".k_conv" in name:
            s = 0.7
        if ".q.weight" in name or ".k.weight" in name:     # autoencoder AttnBlock: make the softmax non-uniform
            s = 3.0 * s
        return hash_tensor(shape, name, s)
    raise KeyError(name)


def filled_state_dict(shapes: Dict[str, Tuple[int, ...]], prefix: str = "") -> Dict[str, torch.Tensor]:
    """{prefix+name: tensor}.  The hash tag is always the UN-prefixed name so the UNet gets the
    same values whether addressed as 'model.enc...' (EDMPrecond) or 'model.model.enc...' (DDPM)."""
    return {prefix + k: fill_value(k, s) for k, s in shapes.items()}
