"""ctypes binding of libadm_hip.so (the C ABI declared in include/adm_hip.h).

There is NO fallback: if the shared library is missing or a kernel rejects its arguments the call
raises.  PyTorch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_double, c_float, c_int, c_long, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadm_hip.so")
_lib = None

P, I, L, F, D, U = c_void_p, c_int, c_long, c_float, c_double, c_uint64
_SIGS = {
    "adm_version": [],
    "adm_conv_fwd": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_splitk": [I, I, I],
    "adm_conv_fwd_ws": [P, P, P, P, P, P, L, I, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_fwd_strided": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_fwd_wino": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_fwd_wino_up": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "adm_pack_weight_wino": [P, P, P, I, I, I, I, P],
    "adm_pack_weight_wino2d": [P, P, P, I, I, I, I, P],
    "adm_conv_fwd_wino2d": [P, P, P, P, P, P, L, I, I, I, I, I, I, I, I, I, P],
    "adm_wino2d_splitk": [I, I, I, I, I],
    "adm_wino2d_variant": [I],
    "adm_wino2d_h3_wide": [I],
    "adm_wgrad_h3_blocks": [I],
    "adm_conv_fwd_wino2d_x6": [P, P, P, P, P, P, L, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_fwd_wino2d_x6_up": [P, P, P, P, P, P, L, I, I, I, I, I, I, I, I, I, P],
    "adm_wino2d_x6_splitk": [I, I, I, I, I],
    "adm_split3_bf16": [P, P, I, I, P],
    "adm_split2_f16": [P, P, I, I, F, P, P],
    "adm_conv_fwd_wino2d_h3": [P, P, P, P, P, P, L, I, I, I, I, I, I, I, I, I, P, F, I, P],
    "adm_gemm_x6": [P, P, P, P, P, L, I, I, I, I, I, I, P],
    "adm_gemm_x6_amax": [P, P, P, P, P, L, I, I, I, I, I, I, P, P],
    "adm_gemm_x6_h3": [P, P, P, P, P, L, I, I, I, I, I, I, P, F, P, P],
    "adm_split2_rows_f16": [P, P, I, I, I, F, P, P],
    "adm_split3_rows": [P, P, I, I, I, P],
    "adm_conv_wgrad_x6": [P, P, P, P, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_x6_up": [P, P, P, P, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_x6_ws": [P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_x6_plan": [I, I, I, I, I],
    "adm_gemm_wgrad_x6": [P, P, P, P, L, I, I, I, I, I, P],
    "adm_gemm_wgrad_x6_ws": [P, P, P, P, L, I, I, I, I, I, P],
    "adm_gemm_wgrad_x6_plan": [L, I, I],
    "adm_conv_wgrad_x6_h3": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, P, P],
    "adm_gemm_wgrad_x6_h3": [P, P, P, P, L, I, I, I, I, I, I, P, P, P],
    "adm_conv_wgrad_x6_bf16a": [P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "adm_gemm_wgrad_x6_bf16a": [P, P, P, P, L, I, I, I, I, I, P],
    "adm_conv_wgrad": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_wino": [P, P, P, P, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_wino_up": [P, P, P, P, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_bias": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_wino2d": [P, P, P, P, I, I, I, I, I, I, I, I, P],
    "adm_unpack_wgrad_wino2d": [P, I, P, I, I, I, I, I, P, P, P],
    "adm_conv_wgrad_plan": [I, I, I, I, I, I, I, I],
    "adm_conv_wgrad_ws": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P],
    "adm_unpack_wgrad_splits": [P, I, P, I, I, I, I, I, I, I, P, P, P],
    "adm_conv_fwd_bf16": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_bf16": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_fwd_bf16a": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "adm_conv_wgrad_bf16a": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "adm_gn_fwd_bf16out": [P, P, P, P, P, P, L, P, I, I, I, I, F, I, F, U, P],
    "adm_f32_to_bf16": [P, P, L, P],
    "adm_pack_weight": [P, P, P, I, I, I, I, I, I, P],
    "adm_pack_weight_table": [P, I, L, P],
    "adm_unpack_wgrad_table": [P, I, L, P],
    "adm_gn_bwd_param_table": [P, I, L, P],
    "adm_unpack_wgrad": [P, P, I, I, I, I, I, I, I, P],
    "adm_permute_vec": [P, P, I, I, I, I, P],
    "adm_colsum": [P, P, I, I, I, I, P],
    "adm_gn_splits": [I, I],
    "adm_gn_fused": [I],
    "adm_gn_stats": [P, P, P, I, I, I, I, F, P],
    "adm_gn_apply": [P, P, P, P, P, L, P, I, I, I, I, I, F, U, P],
    "adm_gn_fwd": [P, P, P, P, P, P, L, P, I, I, I, I, F, I, F, U, P],
    "adm_gn_fwd_amax": [P, P, P, P, P, P, L, P, P, I, I, I, I, F, I, F, U, P],
    "adm_gn_bwd": [P, P, P, P, P, P, L, P, P, P, P, P, I, I, I, I, I, F, U, P],
    "adm_gn_bwd_add": [P, P, P, P, P, P, L, P, P, P, P, P, P, I, I, I, I, I, F, U, P],
    "adm_gn_bwd_add_amax": [P, P, P, P, P, P, L, P, P, P, P, P, P, P, I, I, I, I, I, F, U, P],
    "adm_softmax_rows": [P, L, I, L, F, P],
    "adm_posterior_sample": [P, I, P, P, I, L, I, F, P],
    "adm_attn_fwd": [P, P, P, I, I, I, P],
    "adm_attn_bwd": [P, P, P, P, P, P, I, I, I, P],
    "adm_attn_bwd_amax": [P, P, P, P, P, P, P, I, I, I, P],
    "adm_attn_fwd_h3": [P, P, P, P, I, I, I, P],
    "adm_attn_bwd_h3": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "adm_resample2x": [P, P, I, I, I, I, I, F, I, P],
    "adm_nchw_to_nhwc": [P, I, P, L, P, I, I, I, I, P],
    "adm_nchw_to_nhwc_amax": [P, I, P, L, P, P, I, I, I, I, P],
    "adm_precond_out": [P, I, P, I, P, P, L, P, I, I, I, P],
    "adm_precond_out_bwd": [P, P, L, P, I, I, I, I, P],
    "adm_precond_out_bwd_amax": [P, P, L, P, I, P, I, I, I, P],
    "adm_axpby_b": [P, I, P, P, P, L, P, I, L, P],
    "adm_pos_embedding": [P, P, I, I, P],
    "adm_silu_fwd": [P, P, L, P],
    "adm_silu_bwd": [P, P, P, L, P],
    "adm_add": [P, P, P, L, P],
    "adm_add3": [P, P, P, P, P, L, P],
    "adm_copy_channels": [P, I, I, P, I, I, L, I, F, I, P],
    "adm_concat2": [P, I, P, I, P, L, F, P, P],
    "adm_split2": [P, P, I, P, I, L, F, P],
    "adm_spatial_att_fwd": [P, I, P, P, P, P, I, I, I, P],
    "adm_spatial_att_bwd": [P, I, P, P, P, P, P, P, P, I, I, I, P],
    "adm_q_sample": [P, P, P, P, I, L, I, P],
    "adm_ddm_loss": [P, P, P, P, P, P, P, P, F, I, L, P],
    "adm_ddm_loss_latent": [P, P, P, P, P, P, P, P, P, P, P, F, I, L, I, I, P],
    "adm_sampler_step": [P, P, P, D, D, I, I, D, I, L, P],
    "adm_sampler_step_stochastic": [P, P, P, P, P, P, I, I, D, I, I, L, P],
    "adm_aug_workspace_floats": [I, I, I, I],
    "adm_augment_geometric": [P, P, P, P, P, P, I, I, I, I, P],
    "adm_conv_wgrad_strided": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P],
    "adm_pack_weight_tconv": [P, P, I, I, I, I, I, P],
    "adm_col2im": [P, P, I, I, I, I, I, I, I, I, I, P],
    "adm_ws_fwd": [P, P, P, I, I, F, P],
    "adm_ws_bwd": [P, P, P, P, I, I, I, P],
    "adm_lnc_blocks": [L],
    "adm_lnc_fwd": [P, P, P, L, I, F, P],
    "adm_lnc_bwd": [P, P, P, P, P, P, L, I, F, I, P],
    "adm_bn_blocks": [L],
    "adm_bn_fwd": [P, P, P, P, P, P, P, P, L, I, F, F, I, P],
    "adm_bn_bwd": [P, P, P, P, P, P, P, P, P, L, I, I, I, P],
    "adm_bilinear_fwd": [P, P, I, I, I, I, I, I, I, I, I, P],
    "adm_bilinear_bwd": [P, P, I, I, I, I, I, I, I, I, I, P],
    "adm_act_fwd": [P, P, L, I, F, U, P],
    "adm_act_bwd": [P, P, P, L, I, F, U, P],
    "adm_fourier_features": [P, P, P, I, I, P],
    "adm_spatial_att_big_fwd": [P, I, P, P, P, P, P, I, I, I, P],
    "adm_spatial_att_big_bwd": [P, I, P, P, P, P, P, P, P, P, I, I, I, P],
    "adm_mha_fwd": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, P],
    "adm_mha_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, F, P],
    "adm_linattn_ws_floats": [I, I],
    "adm_linattn_fwd": [P, P, P, P, P, I, I, P],
    "adm_linattn_bwd": [P, P, P, P, P, P, P, P, I, I, P],
    "adm_sumsq_blocks": [L],
    "adm_sumsq": [P, P, P, L, P],
    "adm_adamw_step": [P, P, P, P, P, P, L, F, F, F, F, F, F, I, F, F, P],
}
EXPORTS = tuple(_SIGS)


def build(verbose: bool = False) -> str:
    """Compile libadm_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-8000:])
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libadm_hip.so failed")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no non-HIP fallback for the adm_amd hot path)")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(_lib, name)      # AttributeError if the .so lacks a declared symbol
            fn.argtypes = args
            fn.restype = c_long if name in ("adm_aug_workspace_floats", "adm_linattn_ws_floats") else c_int
        if os.environ.get("ADM_H3_WIDE") is not None:      # A/B switch: form of the fp16-format 3x3 kernel (-1 per launch, 0 / 1)
            _lib.adm_wino2d_h3_wide(int(os.environ["ADM_H3_WIDE"]))
        if os.environ.get("ADM_WGRAD_H3_BLOCKS") is not None:      # A/B switch: 64-cout blocks per workgroup of the fp16-format weight gradient
            _lib.adm_wgrad_h3_blocks(int(os.environ["ADM_WGRAD_H3_BLOCKS"]))
    return _lib


def stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> c_void_p:
    return None if t is None else c_void_p(t.data_ptr())


NO_STREAM = ("adm_version", "adm_conv_splitk", "adm_gn_splits", "adm_aug_workspace_floats", "adm_conv_wgrad_plan",
             "adm_sumsq_blocks", "adm_lnc_blocks", "adm_bn_blocks", "adm_linattn_ws_floats", "adm_wino2d_splitk", "adm_wino2d_x6_splitk", "adm_wino2d_variant", "adm_wino2d_h3_wide", "adm_wgrad_h3_blocks", "adm_gn_fused", "adm_conv_wgrad_x6_plan", "adm_gemm_wgrad_x6_plan")      # host-side queries: no stream argument, called as lib().name(...)


def call(name: str, *args):
    rc = getattr(lib(), name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc} (-22 = unsupported shape/alignment, -5 = launch failure)")


def require_cuda(t: torch.Tensor, what: str = "tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"adm_amd: {what} is on {t.device}; the HIP hot path needs a GPU tensor "
                           "(no CPU fallback exists by design)")
