"""ctypes binding of libn3d.so (include/n3d.h).

The product path has NO fallback: if the library is missing, was not built for gfx950, or a
tensor is not on a HIP device, the call raises.  (tests/ use the CPU oracle only as a checker.)
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own HIP runtime (libamdhip64.so.7); it must be the one already resident when
# libn3d.so is dlopen'ed so that both share ONE runtime (same soname -> the loader reuses it).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("N3D_LIB") or os.path.join(_HERE, "libn3d.so")   # N3D_LIB: another build of the same ABI (A/B timing of a kernel change)


class N3DError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    """n3d_conv_geom (include/n3d.h)"""
    _fields_ = [(n, C.c_int32) for n in
                ("B", "Di", "Hi", "Wi", "Ci", "Do", "Ho", "Wo", "Co", "k", "stride", "dil", "pad", "depthwise")]


class PackJob(C.Structure):
    """n3d_pack_job (include/n3d.h)"""
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p)] + [(n, C.c_int32) for n in ("Co", "Ci", "taps", "data_grad", "layout", "cdp")]


class FinalJob(C.Structure):
    """n3d_final_job (include/n3d.h)"""
    _fields_ = [("partial", C.c_void_p), ("pbias", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("nchunks", "ntiles", "tci", "tco", "ci_t", "co_t", "Co", "Ci", "taps", "pad_")]


class GnFwdTerm(C.Structure):
    """n3d_gn_fwd_term (include/n3d.h)"""
    _fields_ = [("raw", C.c_void_p), ("rld", C.c_int64), ("stats", C.c_void_p), ("rows", C.c_int32), ("relu", C.c_int32),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("wptr", C.c_void_p), ("a_out", C.c_void_p), ("b_out", C.c_void_p),
                ("mean_rstd_out", C.c_void_p), ("sumraw", C.c_void_p), ("dtype", C.c_int32), ("pad_", C.c_int32)]


class GnBwdTerm(C.Structure):
    """n3d_gn_bwd_term (include/n3d.h)"""
    _fields_ = [("raw", C.c_void_p), ("rld", C.c_int64), ("a", C.c_void_p), ("b", C.c_void_p), ("sums", C.c_void_p),
                ("rows", C.c_int32), ("relu", C.c_int32), ("gamma", C.c_void_p), ("mean_rstd", C.c_void_p), ("wptr", C.c_void_p),
                ("sumraw", C.c_void_p), ("draw", C.c_void_p), ("drld", C.c_int64), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("dalpha", C.c_void_p), ("dbias_conv", C.c_void_p), ("cA", C.c_void_p), ("cB", C.c_void_p), ("cC", C.c_void_p),
                ("dtype", C.c_int32), ("pad_", C.c_int32)]


class SeTerm(C.Structure):
    """n3d_se_term (include/n3d.h)"""
    _fields_ = [("sums", C.c_void_p), ("rows", C.c_int32), ("pad_", C.c_int32), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_void_p), ("mean", C.c_void_p), ("hidden", C.c_void_p), ("gate", C.c_void_p), ("wptr", C.c_void_p),
                ("dw1", C.c_void_p), ("db1", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p), ("dalpha", C.c_void_p),
                ("A", C.c_void_p), ("Bc", C.c_void_p)]


class PlainCoefTerm(C.Structure):
    """n3d_plain_coef_term (include/n3d.h)"""
    _fields_ = [("sums", C.c_void_p), ("rows", C.c_int32), ("pad_", C.c_int32), ("wptr", C.c_void_p), ("dalpha", C.c_void_p), ("A", C.c_void_p)]


class DwJob(C.Structure):
    """n3d_dw_job (include/n3d.h)"""
    _fields_ = [("g", C.POINTER(ConvGeom)), ("data_grad", C.c_int32), ("flags", C.c_int32), ("src", C.c_void_p), ("sld", C.c_int64),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("dst", C.c_void_p), ("dld", C.c_int64)]


class ConvFwdCall(C.Structure):
    """n3d_conv_fwd_call (include/n3d.h)"""
    _fields_ = [("g", C.POINTER(ConvGeom)), ("transposed", C.c_int32), ("flags", C.c_int32), ("x", C.c_void_p), ("xld", C.c_int64),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("yld", C.c_int64), ("in_gate", C.c_void_p),
                ("stats", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


class ConvBwdCall(C.Structure):
    """n3d_conv_bwd_call (include/n3d.h)"""
    _fields_ = [("g", C.POINTER(ConvGeom)), ("transposed", C.c_int32), ("flags_data", C.c_int32), ("flags_weight", C.c_int32),
                ("pad_", C.c_int32), ("x", C.c_void_p), ("xld", C.c_int64), ("dy", C.c_void_p), ("dyld", C.c_int64), ("w", C.c_void_p),
                ("dx", C.c_void_p), ("dxld", C.c_int64), ("relu_src", C.c_void_p), ("rld", C.c_int64), ("out_gate", C.c_void_p),
                ("ws_data", C.c_void_p), ("ws_data_bytes", C.c_size_t), ("dw", C.c_void_p), ("dbias", C.c_void_p),
                ("in_gate", C.c_void_p), ("ws_weight", C.c_void_p), ("ws_weight_bytes", C.c_size_t), ("deferred", C.POINTER(FinalJob))]


class Head(C.Structure):
    """n3d_head (include/n3d.h)"""
    _fields_ = [("x", C.c_void_p), ("xld", C.c_int64), ("x_dtype", C.c_int32), ("B", C.c_int32), ("Ci", C.c_int32), ("Co", C.c_int32),
                ("N", C.c_int64), ("w", C.c_void_p), ("bias", C.c_void_p), ("gate", C.c_void_p),
                ("x_node_stride", C.c_int64), ("dx_node_stride", C.c_int64), ("node_c", C.c_int32), ("t_dtype", C.c_int32)]


class PatchDesc(C.Structure):
    """n3d_patch_desc (include/n3d.h)"""
    _fields_ = [("corner", C.c_int32 * 3), ("perm", C.c_int32 * 3), ("flip", C.c_int32 * 3)]


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float
_sz = C.c_size_t
_gp = C.POINTER(ConvGeom)

# name -> (restype, argtypes); must list every function declared in include/n3d.h
PROTOTYPES = {
    "n3d_last_error": (C.c_char_p, []),
    "n3d_version": (_i, []),
    "n3d_device_ok": (_i, []),
    "n3d_zero": (_i, [_p, _sz, _p]),
    "n3d_conv_workspace_bytes": (_sz, [_gp]),
    "n3d_conv_stats_rows": (_i, [_gp, _i, _i]),
    "n3d_stats_rows": (_i, [_i64, _i]),
    "n3d_conv_fwd": (_i, [_gp, _p, _i64, _p, _p, _p, _i64, _i, _p, _p, _p, _sz, _p]),
    "n3d_conv_k1_norm_ok": (_i, [_gp]),
    "n3d_conv_k1_norm_rows": (_i, [_gp]),
    "n3d_conv_k1_norm_fwd": (_i, [_gp, _p, _i64, _p, _p, _p, _i64, _i, _p, _p, _p, _p, _sz, _p]),
    "n3d_conv_k1_norm_bwd_reduce": (_i, [_gp, _p, _i64, _p, _p, _p, _i64, _p, _p, _i, _p, _p]),
    "n3d_conv_k1_norm_bwd_apply_wgrad": (_i, [_gp, _p, _i64, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i, _p, _p, _sz, C.POINTER(FinalJob), _p]),
    "n3d_conv_bwd_data": (_i, [_gp, _p, _i64, _p, _p, _i64, _i, _p, _i64, _p, _p, _sz, _p]),
    "n3d_conv_bwd_weight": (_i, [_gp, _p, _i64, _p, _i64, _p, _p, _i, _p, _p, _sz, C.POINTER(FinalJob), _p]),
    "n3d_conv_fwd2": (_i, [C.POINTER(ConvFwdCall), C.POINTER(ConvFwdCall), _p]),
    "n3d_conv_fwdN": (_i, [C.POINTER(ConvFwdCall), _i, _p]),
    "n3d_conv_bwd_both2": (_i, [C.POINTER(ConvBwdCall), C.POINTER(ConvBwdCall), _p]),
    "n3d_conv_bwd_data2": (_i, [C.POINTER(ConvBwdCall), C.POINTER(ConvBwdCall), _p]),
    "n3d_convT_bwd_both": (_i, [_gp, _p, _i64, _p, _i64, _p, _p, _i64, _i, _p, _sz, _p, _i, _p, _sz, C.POINTER(FinalJob), _p]),
    "n3d_conv_bwd_both": (_i, [_gp, _p, _i64, _p, _i64, _p, _p, _i64, _i, _p, _i64, _p, _p, _sz, _p, _p, _i, _p, _p, _sz,
                                C.POINTER(FinalJob), _p]),
    "n3d_conv_pack_info": (_i, [_gp, _i, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "n3d_pack_batch": (_i, [C.POINTER(PackJob), _i, _p]),
    "n3d_wgrad_finalize_batch": (_i, [C.POINTER(FinalJob), _i, _p]),
    "n3d_selftest_job_tables": (_i, []),
    "n3d_convT_fwd": (_i, [_gp, _p, _i64, _p, _p, _p, _i64, _i, _p, _p, _p, _sz, _p]),
    "n3d_convT_bwd_data": (_i, [_gp, _p, _i64, _p, _p, _i64, _i, _p, _sz, _p]),
    "n3d_convT_bwd_weight": (_i, [_gp, _p, _i64, _p, _i64, _p, _p, _i, _p, _sz, C.POINTER(FinalJob), _p]),
    "n3d_channel_stats": (_i, [_p, _i64, _i, _i64, _i, _p, _p]),
    "n3d_channel_stats_t": (_i, [_p, _i64, _i, _i, _i64, _i, _p, _p]),
    "n3d_gn_coeffs": (_i, [_p, _i, _p, _p, _i, _i, _i, _i64, _f, _p, _p, _p, _p, _p]),
    "n3d_affine_act": (_i, [_p, _i64, _p, _p, _p, _p, _i64, _i, _i64, _i, _i, _p]),
    "n3d_fused_max_rows": (_i, []),
    "n3d_affine_act_gn": (_i, [_p, _i64, _p, _i, _p, _p, _i, _f, _p, _p, _i64, _i, _i64, _i, _i, _p, _p, _p, _p, _p]),
    "n3d_affine_act_bwd_apply_gn": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i64, _i, _i64, _i, _i, _i,
                                         _p, _p, _p, _p, _p]),
    "n3d_gn_coeffs2": (_i, [C.POINTER(GnFwdTerm), C.POINTER(GnFwdTerm), _i, _i, _i, _i64, _f, _p]),
    "n3d_affine_act2": (_i, [C.POINTER(GnFwdTerm), C.POINTER(GnFwdTerm), _p, _i64, _p, _i64, _i, _i64, _i, _i, _p]),
    "n3d_bwd_small2_ok": (_i, [_i, _i64, _i, _i]),
    "n3d_affine_act_bwd_small2": (_i, [_p, _i64, _p, _i64, C.POINTER(GnBwdTerm), C.POINTER(GnBwdTerm), _i, _i64, _i, _i, _p]),
    "n3d_bwd_small_mode": (_i, [_i, _i64, _i, _i]),
    "n3d_bwd_small_scratch_bytes": (C.c_size_t, [_i, _i]),
    "n3d_affine_act_bwd_small": (_i, [_p, _i64, _p, _i64, C.POINTER(GnBwdTerm), C.POINTER(GnBwdTerm), _i, _i64, _i, _i, _p, C.c_size_t, _p, _p]),
    "n3d_gn_bwd_coeffs2": (_i, [C.POINTER(GnBwdTerm), C.POINTER(GnBwdTerm), _i, _i, _i, _i64, _p]),
    "n3d_affine_act_bwd_apply2": (_i, [_p, _i64, _p, _i64, C.POINTER(GnBwdTerm), C.POINTER(GnBwdTerm), _i, _i64, _i, _p]),
    "n3d_plain_bwd_coeffsN": (_i, [C.POINTER(PlainCoefTerm), _i, _i, _i, _p]),
    "n3d_pool2_fwd_both": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "n3d_pool2_bwd_both": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "n3d_pool2_bwd_scaled": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _i, _i, _i, _p, _p]),
    "n3d_dwconv_batch": (_i, [C.POINTER(DwJob), _i, _p]),
    "n3d_se_gate_fwdN": (_i, [C.POINTER(SeTerm), _i, _i64, _i, _i, _p]),
    "n3d_se_gate_bwdN": (_i, [C.POINTER(SeTerm), _i, _i64, _i, _i, _p]),
    "n3d_node_bwd_coeffs": (_i, [_p, _i, _p, _i, _i, _i, _i, _i64, _p]),
    "n3d_affine_act_bwd_apply_sum": (_i, [_p, _i64, _p, _i, _i, _i64, _i, _p]),
    "n3d_node_fwd_coeffs": (_i, [_p, _i, _p, _i, _i, _i, _i, _i64, _f, _p]),
    "n3d_channel_statsN": (_i, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), _i, _i, _i64, _i, _p]),
    "n3d_gn_coeffsN": (_i, [C.POINTER(GnFwdTerm), _i, _i, _i, _i, _i64, C.c_float, _p]),
    "n3d_affine_actN": (_i, [C.POINTER(GnFwdTerm), _i, _p, _i64, _i, _i64, _i, _i, _p]),
    "n3d_affine_act_bwd_reduceN": (_i, [_p, _i64, C.POINTER(GnBwdTerm), _i, _i, _i64, _i, _p]),
    "n3d_gn_bwd_coeffsN": (_i, [C.POINTER(GnBwdTerm), _i, _i, _i, _i, _i64, _p]),
    "n3d_affine_act_bwd_applyN": (_i, [_p, _i64, C.POINTER(GnBwdTerm), _i, _i, _i64, _i, _p]),
    "n3d_affine_act_gn2": (_i, [C.POINTER(GnFwdTerm), C.POINTER(GnFwdTerm), _i, _f, _p, _i64, _p, _i64, _i, _i64, _i, _i, _p]),
    "n3d_affine_act_bwd_reduce2": (_i, [_p, _i64, _p, _i64, C.POINTER(GnBwdTerm), C.POINTER(GnBwdTerm), _i, _i64, _i, _p]),
    "n3d_affine_act_bwd_apply_gn2": (_i, [_p, _i64, _p, _i64, C.POINTER(GnBwdTerm), C.POINTER(GnBwdTerm), _i, _i64, _i, _i, _p]),
    "n3d_affine_act_bwd_reduce": (_i, [_p, _i64, _p, _i64, _p, _p, _i, _i64, _i, _i, _p, _p]),
    "n3d_gn_bwd_coeffs": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "n3d_plain_bwd_coeffs": (_i, [_p, _i, _p, _i, _i, _p, _p, _p]),
    "n3d_affine_act_bwd_apply": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i, _i64, _i, _i, _p]),
    "n3d_se_gate_fwd": (_i, [_p, _i, _i64, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "n3d_se_gate_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "n3d_pool2_fwd": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _i, _i, _i, _p]),
    "n3d_pool2_bwd": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _i, _i, _i, _p]),
    "n3d_dice_rows": (_i, [_i64]),
    "n3d_dice_fwd": (_i, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i, _i, _i64, _f, _p, _p, _p, _p]),
    "n3d_dice_bwd": (_i, [_p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _i, _i, _i64, _f, _p, _p, _p, _i64, _i64, _i64, _p]),
    "n3d_dropout3d_uniform": (C.c_float, [C.c_uint64, C.c_uint32, C.c_uint32]),
    "n3d_dropout3d_gate": (_i, [_p, _f, _i, _i, _p, _p]),
    "n3d_head_rows": (_i, [_i64]),
    "n3d_head_workspace_bytes": (_sz, [C.POINTER(Head)]),
    "n3d_head_fwd": (_i, [C.POINTER(Head), _p, _i64, _i64, _i64, _p, _p, _i64, _i64, _i64, _f, _p, _p, _p, _p]),
    "n3d_head_bwd": (_i, [C.POINTER(Head), _p, _i64, _i64, _i64, _p, _i64, _i64, _i64, _f, _p, _p, _p, _i64, _i, _i, _p, _p, _p, _sz,
                          C.POINTER(FinalJob), _p]),
    "n3d_ncdhw_to_ndhwc": (_i, [_p, _p, _i64, _i, _i, _i64, _p]),
    "n3d_ndhwc_to_ncdhw": (_i, [_p, _i64, _p, _i, _i, _i64, _p]),
    "n3d_patch_batch": (_i, [_p, _i, _p, _i, _i, _i, C.POINTER(PatchDesc), _i, _i, _i, _p, _i64, _p, _p]),
    "n3d_stitch": (_i, [_p, _i64, _i64, _i64, _i, _i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "n3d_tumor_labels": (_i, [_p, _i64, C.c_double, _i, _p, _p]),
    "n3d_comm_available": (_i, []),
    "n3d_comm_unique_id": (_i, [_p]),
    "n3d_comm_init": (_i, [_p, _i, _i, C.POINTER(C.c_void_p)]),
    "n3d_comm_allreduce_sum": (_i, [_p, _p, _i64, _p]),
    "n3d_comm_broadcast": (_i, [_p, _p, _i64, _i, _p]),
    "n3d_comm_destroy": (_i, [_p]),
    "n3d_adam_step": (_i, [_p, _p, _p, _p, _i64, _f, _p, _f, _f, _f, _f, _f, _p, _i, _p]),
    "n3d_adam_step_guarded": (_i, [_p, _p, _p, _p, _i64, _f, _p, _f, _f, _f, _f, _f, _p, _i, _p, _p, _p, _p, _p, _p]),
    "n3d_guard_flag": (_i, [_p, _p, _p, _p]),
    "n3d_host_word_alloc": (_i, [C.POINTER(C.c_void_p)]),
    "n3d_host_word_free": (_i, [_p]),
    "n3d_stream_create_low_priority": (_i, [C.POINTER(C.c_void_p)]),
    "n3d_stream_capture_begin": (_i, [_p]),
    "n3d_stream_capture_end": (_i, [_p, C.POINTER(C.c_void_p)]),
    "n3d_graph_launch": (_i, [_p, _p]),
    "n3d_graph_destroy": (_i, [_p]),
    "n3d_sync_signal": (_i, [_p, _p, _i, _p]),
    "n3d_sync_wait": (_i, [_p, _p, _p, _i, _i64, _p]),
    "n3d_sync_wait2": (_i, [_p, _p, _p, _p, _i, _i64, _p]),
    "n3d_stamp": (_i, [_p, _p]),
}

# flags (include/n3d.h)
RELU_IN, RELU, ACCUMULATE, POOL_MAX, NO_MFMA, PREPACKED = 1, 2, 4, 8, 16, 32
F32, BF16, U8 = 0, 1, 2   # N3D_F32 / N3D_BF16 / N3D_U8 (byte targets of the head passes and of the data step)
PATCH_INCLUSIVE, PATCH_T_U8 = 1, 2   # n3d_patch_batch flags
SRC_BF16, DST_BF16, ACT_BF16 = 64, 128, 64   # storage flags of the conv / epilogue families
MM_BF16 = 256   # conv family, bf16 configuration: the C >= 16 MFMA kernels round their operands to bf16 (fp32 storage, fp32 accumulate)

_lib = None


def load(path: str | None = None):
    """dlopen libn3d.so and attach prototypes.  Raises N3DError if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise N3DError(
            "libn3d.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C nas_3d_unet_amd/csrc`; there is no CPU fallback." % p)
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().n3d_last_error()
        raise N3DError("%s failed (%d): %s" % (what or "libn3d call", status, msg.decode() if msg else "?"))


def require_device():
    lib = load()
    if not lib.n3d_device_ok():
        raise N3DError("libn3d needs a gfx950 HIP device: %s" % lib.n3d_last_error().decode())
