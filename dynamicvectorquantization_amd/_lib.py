"""ctypes binding of libdvq_hip.so (include/dvq_hip.h).  There is no CPU fallback: if the library
is missing the product path fails loudly here."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DVQ_USE_PROBES_LIB=1 (tools/debug/ only): the -DDVQ_PROBES build with the wrong-result timing experiments (build.py --probes)
LIB_PATH = os.path.join(HERE, "libdvq_hip_probes.so" if os.environ.get("DVQ_USE_PROBES_LIB", "0") == "1" else "libdvq_hip.so")

F32, BF16 = 0, 1

vp, i64, i32, f32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [("N", i64), ("H", i64), ("W", i64), ("Cin", i64), ("OH", i64), ("OW", i64), ("Cout", i64),
                ("KH", i32), ("KW", i32), ("stride", i32), ("pad_t", i32), ("pad_l", i32), ("upsample", i32),
                ("dtype", i32), ("impl", i32)]


class DecodeLayer(C.Structure):
    """dvq_decode_layer of include/dvq_hip.h"""
    _fields_ = [(n, vp) for n in ("wq", "wk", "wv", "wo", "w1", "w2", "bq", "bk", "bv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b",
                                  "kcache", "vcache")]


class LinPackEntry(C.Structure):
    """LinPackEntry of csrc/misc.hip (dvq_linear_pack_multi)"""
    _fields_ = [("master", vp), ("w", vp), ("wt", vp), ("out", i64), ("in_", i64), ("out_p", i64), ("tile_begin", i64),
                ("wt_ld", i64), ("bias_src", vp), ("bias_dst", vp)]


class PackEntry(C.Structure):
    _fields_ = [("master", vp), ("w", vp), ("wt", vp), ("Cout", i64), ("Cin", i64), ("taps", i64), ("Cin_p", i64),
                ("Cout_p", i64), ("begin", i64), ("dtype", i64)]


# name -> (restype, argtypes); mirrors include/dvq_hip.h one to one
SIGNATURES = {
    "dvq_last_error": (C.c_char_p, []),
    "dvq_version": (i32, []),
    "dvq_check_device": (i32, []),
    "dvq_set_deterministic": (i32, [i32]),
    "dvq_deterministic": (i32, []),
    "dvq_set_fp32_split": (i32, [i32]),
    "dvq_fp32_split": (i32, []),
    "dvq_probe_mfma_rate": (i32, [i32, vp, vp, vp]),
    "dvq_vq_prep_bytes": (sz, [i64, i64]),
    "dvq_vq_prepare": (i32, [vp, i64, i64, vp, vp]),
    "dvq_vq_argmin_workspace_bytes": (sz, [i64]),
    "dvq_vq_argmin": (i32, [vp, i32, vp, vp, i64, i64, i64, vp, vp, i32, vp]),
    "dvq_vq_distances": (i32, [vp, i32, vp, i64, i64, i64, vp, vp]),
    "dvq_vq_gather_loss": (i32, [vp, i32, vp, vp, vp, i64, i64, vp, vp, vp]),
    "dvq_vq_backward": (i32, [vp, vp, i32, vp, vp, vp, vp, i64, i64, vp, vp]),
    "dvq_vq_embed": (i32, [vp, vp, i64, i64, i32, vp, vp]),
    "dvq_vq_ema_stats": (i32, [vp, i32, vp, i64, i64, i64, vp, vp]),
    "dvq_vq_ema_apply": (i32, [vp, vp, f32, f32, i64, i64, vp, vp, vp, vp, vp]),
    "dvq_patch_entropy_gate": (i32, [vp, i64, i64, i64, i32, f32, vp, vp, vp]),
    "dvq_patch_entropy_gate_range": (i32, [vp, i64, i64, i64, i32, f32, f32, f32, vp, vp, vp]),
    "dvq_gn_stats": (i32, [vp, i32, i64, i64, i64, i32, vp, vp]),
    "dvq_gn_apply": (i32, [vp, i32, i64, i64, i64, i32, f32, vp, vp, vp, i32, vp, vp, vp]),
    "dvq_gn_bwd_partial_bytes": (sz, [i64, i64, i64]),
    "dvq_gn_bwd_reduce": (i32, [vp, vp, i32, i64, i64, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
    "dvq_gn_bwd_dx": (i32, [vp, vp, i32, i64, i64, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
    "dvq_conv2d_fwd": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]),
    "dvq_conv3x3_fused_ok": (i32, [C.POINTER(ConvDesc)]),
    "dvq_conv2d_fwd_ex": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "dvq_conv2d_wgrad_oihw_ex": (i32, [C.POINTER(ConvDesc), vp, vp, i64, i64, vp, vp, i32, vp, vp]),
    "dvq_split_bf16_planes": (i32, [vp, vp, vp, i64, i64, i64, vp]),
    "dvq_conv3x3_x3_ok": (i32, [C.POINTER(ConvDesc), i32]),
    "dvq_conv3x3_x3_scratch_bytes": (i64, [C.POINTER(ConvDesc), i32]),
    "dvq_conv2d_fwd_x3": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32, vp, i64, vp]),
    "dvq_conv2d_dgrad_x3": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32, vp, i64, vp]),
    "dvq_conv2d_wgrad_x3_scratch_bytes": (i64, [C.POINTER(ConvDesc)]),
    "dvq_conv2d_wgrad_oihw_x3": (i32, [C.POINTER(ConvDesc), vp, vp, i64, i64, vp, vp, i32, vp, i64, vp]),
    "dvq_gn_scale_shift": (i32, [vp, vp, vp, i64, i64, i64, i32, f32, vp, vp, vp]),
    "dvq_conv2d_dgrad": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp]),
    "dvq_conv2d_fwd_act": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, i32, vp]),
    "dvq_conv2d_dgrad_mask": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i32, vp]),
    "dvq_set_workspace": (i32, [vp, i64]),
    "dvq_workspace_release": (i32, [vp]),
    "dvq_cmdlist_create": (i32, [vp, C.POINTER(vp)]),
    "dvq_cmdlist_replay": (i32, [vp, vp, vp]),
    "dvq_cmdlist_info": (i32, [vp, vp]),
    "dvq_cmdlist_destroy": (i32, [vp]),
    "dvq_halo_trace_read": (i32, [vp, i64]),
    "dvq_permute_dual": (i32, [vp, vp, i64, i32, i32, i32, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp]),
    "dvq_permute_dual_back": (i32, [vp, vp, vp, vp, i64, i64, i64, i32, i32, i64, i64, vp, vp]),
    "dvq_avgpool_slice": (i32, [vp, i32, i64, i64, i64, i64, i32, vp, i64, i64, vp]),
    "dvq_avgpool_slice_bwd": (i32, [vp, i32, i64, i64, i64, i64, i64, i64, i32, vp, vp]),
    "dvq_silu": (i32, [vp, i32, i64, vp, vp]),
    "dvq_silu_bwd": (i32, [vp, vp, i32, i64, vp, vp]),
    "dvq_relu": (i32, [vp, i32, i64, vp, vp]),
    "dvq_relu_bwd": (i32, [vp, vp, i32, i64, vp, vp]),
    "dvq_upsample_nearest2x": (i32, [vp, i32, i64, i64, i64, i64, vp, vp]),
    "dvq_upsample_nearest2x_bwd": (i32, [vp, i32, i64, i64, i64, i64, vp, vp]),
    "dvq_grain_merge": (i32, [vp, i32, vp, vp, i32, i64, i64, i64, i64, vp, vp, vp]),
    "dvq_grain_merge_bwd": (i32, [vp, vp, i32, vp, vp, i32, i64, i64, i64, i64, vp, vp, vp]),
    "dvq_affine_channels": (i32, [vp, i32, i64, i64, vp, vp, vp, vp]),
    "dvq_axpy_dev": (i32, [vp, vp, vp, i32, i64, vp, vp]),
    "dvq_maxpool2x2": (i32, [vp, i32, i64, i64, i64, i64, vp, vp]),
    "dvq_maxpool2x2_relu_bwd": (i32, [vp, vp, vp, i32, i64, i64, i64, i64, vp, vp]),
    "dvq_lpips_head": (i32, [vp, vp, vp, i32, i64, i64, i64, vp, f32, vp, vp]),
    "dvq_lpips_head_drop": (i32, [vp, vp, vp, i32, i64, i64, i64, vp, f32, vp, f32, C.c_uint64, vp]),
    "dvq_conv2d_wgrad": (i32, [C.POINTER(ConvDesc), vp, vp, vp, vp, vp]),
    "dvq_conv2d_wgrad_oihw": (i32, [C.POINTER(ConvDesc), vp, vp, i64, i64, vp, vp, i32, vp]),
    "dvq_pack_weights_multi": (i32, [vp, i64, i64, vp]),
    "dvq_linear_pack_multi": (i32, [vp, i64, i64, vp]),
    "dvq_pack_weight": (i32, [vp, i64, i64, i64, i64, i64, i64, i32, vp, vp, vp]),
    "dvq_unpack_wgrad": (i32, [vp, i64, i64, i64, i64, i64, vp, vp]),
    "dvq_nchw_to_nhwc_pad": (i32, [vp, i64, i64, i64, i64, i64, i32, vp, vp]),
    "dvq_nhwc_pad_to_nchw": (i32, [vp, i32, i64, i64, i64, i64, i64, vp, vp]),
    "dvq_gemm_nt": (i32, [vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, f32, vp, i32, i32, vp]),
    "dvq_gemm_nt_res": (i32, [vp, vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, f32, vp, i32, vp]),
    "dvq_gemm_tn": (i32, [vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, i64, i64, i64, i64, i32, vp]),
    "dvq_gemm_tn_colsum": (i32, [vp, vp, vp, vp, i32, i64, i64, i64, i64, i64, i64, i32, vp]),
    "dvq_softmax_rows": (i32, [vp, i32, i64, i64, f32, vp, vp]),
    "dvq_softmax_rows_bwd": (i32, [vp, vp, i32, i64, i64, f32, vp, vp]),
    "dvq_transpose": (i32, [vp, i32, i64, i64, i64, vp, vp]),
    "dvq_dual_merge": (i32, [vp, vp, vp, i32, i64, i64, i64, i64, vp, vp, vp]),
    "dvq_dual_merge_bwd": (i32, [vp, vp, i32, i64, i64, i64, i64, vp, vp, vp]),
    "dvq_add": (i32, [vp, vp, i32, i64, vp, vp]),
    "dvq_add_bias_bcast": (i32, [vp, vp, i32, i64, i64, vp, vp]),
    "dvq_channel_shift_add8": (i32, [vp, vp, i32, i64, i32, vp, vp]),
    "dvq_sum_batch": (i32, [vp, i32, i64, i64, vp, vp]),
    "dvq_sumpool2x2": (i32, [vp, i32, i64, i64, i64, i64, vp, vp]),
    "dvq_cast": (i32, [vp, i32, vp, i32, i64, vp]),
    "dvq_l1_loss": (i32, [vp, vp, i64, vp, vp, vp, vp]),
    "dvq_adam": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, vp]),
    "dvq_adamw": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp]),
    "dvq_layernorm_fwd": (i32, [vp, i32, i64, i64, f32, vp, vp, vp, vp, vp]),
    "dvq_layernorm_bwd": (i32, [vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, vp]),
    "dvq_layernorm_bwd_res": (i32, [vp, vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, vp]),
    "dvq_layernorm_bwd_res_drop": (i32, [vp, vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, vp, f32, C.c_uint64, vp]),
    "dvq_gelu": (i32, [vp, i32, i64, vp, vp]),
    "dvq_gelu_bwd": (i32, [vp, vp, i32, i64, vp, vp]),
    "dvq_softmax_causal": (i32, [vp, i32, i64, i64, i64, i64, f32, vp, vp]),
    "dvq_embed_gather": (i32, [vp, i64, vp, i32, i64, i64, i64, i64, i64, i32, vp, vp]),
    "dvq_embed_scatter_add": (i32, [vp, i64, vp, i32, i64, i64, i64, i64, i64, i64, i64, vp, vp]),
    "dvq_cross_entropy": (i32, [vp, i32, i64, i64, i64, vp, i64, vp, vp, vp, vp, vp]),
    "dvq_attn_decode": (i32, [vp, vp, vp, i32, i64, i64, i64, i64, i64, f32, vp, vp]),
    "dvq_attn_causal_scratch_bytes": (i64, [i64, i64, i32, i32, i32]),
    "dvq_attn_causal_mask_bytes": (i64, [i64, i64, i32]),
    "dvq_attn_causal_fwd": (i32, [vp, vp, vp, i32, i64, i64, i32, i32, f32, f32, C.c_uint64, vp, vp, vp, vp, vp]),
    "dvq_attn_causal_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i64, i32, i32, f32, f32, C.c_uint64, vp, vp, vp, vp, vp, vp]),
    "dvq_attn_causal_fwd_ld": (i32, [vp, vp, vp, i64, i32, i64, i64, i32, i32, f32, f32, C.c_uint64, vp, vp, vp, vp]),
    "dvq_attn_causal_bwd_ld": (i32, [vp, vp, vp, i64, vp, vp, vp, i32, i64, i64, i32, i32, f32, f32, C.c_uint64, vp, vp, vp, vp, vp, vp]),
    "dvq_attn_full_scratch_bytes": (i64, [i64, i64, i32, i32]),
    "dvq_attn_full_fwd": (i32, [vp, vp, vp, i32, i64, i64, i32, f32, vp, vp, vp, vp]),
    "dvq_attn_full_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i64, i32, f32, vp, vp, vp, vp, vp]),
    "dvq_attn_decode_dev": (i32, [vp, vp, vp, vp, vp, i32, i64, i64, i64, vp, i64, f32, vp, vp]),
    "dvq_rows_dev": (i32, [vp, vp, i32, i64, i64, i64, vp, i32, vp]),
    "dvq_decode_stack_scratch_bytes": (sz, [i64, i64, i64]),
    "dvq_decode_stack_status": (i32, [vp, i64, i64, i64, i32, vp]),
    "dvq_decode_stack": (i32, [vp, i32, i64, i64, i32, i64, i64, vp, C.c_float, vp, vp, i32, vp, vp]),
    "dvq_dropout": (i32, [vp, i32, i64, f32, C.c_uint64, vp, vp]),
    "dvq_dropout_add": (i32, [vp, vp, i32, i64, f32, C.c_uint64, vp, vp]),
    "dvq_fill_f32": (i32, [vp, f32, i64, vp]),
    "dvq_image_desc_bytes": (sz, []),
    "dvq_image_batch_transform": (i32, [vp, vp, vp, vp, i64, i32, i32, vp, vp]),
    "dvq_adamw_dev": (i32, [vp, vp, vp, vp, i64, vp, vp]),
    "dvq_set_f32x8": (i32, [vp, f32, f32, f32, f32, f32, f32, f32, f32, vp]),
    "dvq_sample_rows": (i32, [vp, i64, i64, vp, vp]),
    "dvq_add_uniform": (i32, [vp, i64, f32, vp, vp]),
    "dvq_sample_constrained": (i32, [vp, i32, i64, i64, i64, f32, vp, i64, i64, i64, vp, i64, i64, i64, vp, i32, f32, i32, vp, vp, vp]),
}


class DvqError(RuntimeError):
    pass


_lib = None


def load():
    """Load libdvq_hip.so and declare every signature.  Raises DvqError if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process (streams and
    # device pointers come from torch), so load torch BEFORE libdvq_hip.so resolves its dependency
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DvqError(
            f"{LIB_PATH} not found. Build it with `python -m dynamicvectorquantization_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the DQ-VAE hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so is stale w.r.t. the header
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_launch_hook = None          # debugging (runtime.StepGraph, DVQ_GRAPH_DEBUG): called with the entry-point name after every call


def check(rc: int, what: str = ""):
    if _launch_hook is not None:
        _launch_hook(what)
    if rc != 0:
        msg = load().dvq_last_error().decode("utf-8", "replace")
        raise DvqError(f"{what or 'libdvq_hip'} failed (code {rc}): {msg}")
