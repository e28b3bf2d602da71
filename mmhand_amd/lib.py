"""ctypes binding of libmmhand_hip.so (the C-ABI declared in include/mmhand_hip.h).

The product path has no fallback: if the shared library is missing or a call
returns non-zero, a RuntimeError is raised.  Nothing under ``oracle/`` is ever
imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMH_LIB_PATH: another BUILD of the same library for A/B tools (make AB=1 ... OUT=../libmmhand_hip_ab.so adds the kernels that
# lost their comparison); never a different implementation, and a missing file is an error like the default one
LIB_PATH = os.environ.get("MMH_LIB_PATH") or os.path.join(_HERE, "libmmhand_hip.so")

PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
F32, BF16, FP16 = 0, 1, 2


class ConvDesc(C.Structure):
    """mirror of mmh_conv_desc"""
    _fields_ = [(n, C.c_int32) for n in (
        "B", "H", "W", "Cin", "Cout", "kh", "kw", "stride", "pad", "pad_mode",
        "Ho", "Wo", "x_cs", "y_cs", "dtype")]


class PlaneSrc(C.Structure):
    """mirror of mmh_plane_src"""
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32),
                ("sb", C.c_int64), ("sc", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)]


_vp, _i, _i64, _f, _d, _u64, _sz = (C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double,
                                    C.c_uint64, C.c_size_t)
_DP = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol declared in include/mmhand_hip.h
SIGNATURES = {
    "mmh_last_error": (C.c_char_p, []),
    "mmh_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _vp, _f, _vp, _vp, _vp, _vp]),
    "mmh_set_dropout_salt": (_i, [_vp]),
    "mmh_lp16_clock_stamps": (_i, [_vp, _i]),
    "mmh_u64_add": (_i, [_vp, _u64, _vp]),
    "mmh_pool_exchange": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp]),
    "mmh_version": (_i, []),
    "mmh_set_option": (_i, [C.c_char_p, _i]),
    "mmh_conv2d_fprop": (_i, [_DP, _vp, _vp, _vp, _vp, _i, _vp]),
    "mmh_conv2d_fprop_stats_chunks": (_i, [_DP]),
    "mmh_conv2d_fprop_stats": (_i, [_DP, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmh_conv2d_dgrad": (_i, [_DP, _vp, _vp, _vp, _i, _vp]),
    "mmh_conv2d_dgrad_folded_ws_bytes": (_sz, [_DP]),
    "mmh_conv2d_dgrad_folded": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mmh_wino_weights": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mmh_wino_input": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mmh_wino_dy": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mmh_wino_input_dy": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mmh_wino_gemm": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "mmh_wino_gemm_levels": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "mmh_wino_weights_multi": (_i, [_vp, _i, _i64, _vp]),
    "mmh_prep_weights_lp16_multi": (_i, [_vp, _i, _i64, _vp]),
    "mmh_wino_output": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "mmh_norm_stats_merge": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mmh_norm_stats_merge2_ws_bytes": (_sz, [_i, _i]),
    "mmh_norm_stats_merge2": (_i, [_vp, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "mmh_norm_stats_merge_finalize": (_i, [_vp, _i, _i, _i, _d, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmh_wino_wgrad_gemm_ws_bytes": (_sz, [_i64, _i, _i, _i]),
    "mmh_wino_wgrad_gemm": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "mmh_wino_dw": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "mmh_conv2d_dgrad_border_ws_bytes": (_sz, [_DP]),
    "mmh_conv2d_dgrad_border": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "mmh_conv7_thin_fprop": (_i, [_DP, _vp, _vp, _vp, _vp, _i, _vp]),
    "mmh_conv7_thin_dgrad_ws_bytes": (_sz, [_DP]),
    "mmh_conv7_thin_dgrad": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "mmh_conv7_thin_wgrad_ws_bytes": (_sz, [_DP]),
    "mmh_conv7_thin_wgrad": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "mmh_conv7_stem_wgrad_supported": (_i, [_DP]),
    "mmh_conv7_stem_wgrad_ws_bytes": (_sz, [_DP]),
    "mmh_conv7_stem_wgrad": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "mmh_conv2d_wgrad_ws_bytes": (_sz, [_DP]),
    "mmh_conv2d_wgrad": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "mmh_convT2d_fprop": (_i, [_DP, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mmh_convT2d_dgrad": (_i, [_DP, _vp, _vp, _vp, _vp]),
    "mmh_convT2d_wgrad": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "mmh_reflect_fold": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mmh_prep_weights_bf16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mmh_prep_weights_bf16_flat": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mmh_prep_weights_fp16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mmh_prep_weights_fp16_flat": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mmh_cvt_lp16": (_i, [_vp, _i64, _i, _vp, _vp]),
    "mmh_conv_lp16_supported": (_i, [_DP, _i]),
    "mmh_conv_lp16": (_i, [_DP, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mmh_conv_lp16_stats_chunks": (_i, [_DP]),
    "mmh_conv_lp16_fprop_stats": (_i, [_DP, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmh_conv_stem16_stats_chunks": (_i, [_DP, _i]),
    "mmh_conv_stem16_stats": (_i, [_DP, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmh_lp16_pad_cvt": (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    "mmh_prep_weights_lp16_flat8": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mmh_conv_lp16_flat_supported": (_i, [_DP, _i]),
    "mmh_conv_lp16_flat": (_i, [_DP, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mmh_conv_stem16_supported": (_i, [_DP, _i]),
    "mmh_dgrad_s2_halo_supported": (_i, [_DP, _i]),
    "mmh_conv_stem16_weights_bytes": (_sz, [_i]),
    "mmh_prep_weights_stem16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "mmh_conv_stem16_weights_bytes_k": (_sz, [_i, _i]),
    "mmh_prep_weights_stem16_k": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mmh_conv_stem16": (_i, [_DP, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mmh_wgrad_stem_lp16_supported": (_i, [_DP, _i]),
    "mmh_wgrad_stem_lp16_ws_bytes": (_sz, [_DP, _i]),
    "mmh_wgrad_stem_lp16": (_i, [_DP, _vp, _i, _vp, _vp, _vp, _sz, _i, _vp, _vp]),
    "mmh_conv7_head_wgrad_lp16_supported": (_i, [_DP]),
    "mmh_conv7_head_wgrad_lp16_ws_bytes": (_sz, [_DP]),
    "mmh_conv7_head_wgrad_lp16": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _vp, _vp]),
    "mmh_wgrad_lp16_flat_supported": (_i, [_DP, _i]),
    "mmh_wgrad_lp16_flat_ws_bytes": (_sz, [_DP, _i]),
    "mmh_wgrad_lp16_flat": (_i, [_DP, _vp, _i, _i, _vp, _vp, _vp, _sz, _i, _vp, _vp]),
    "mmh_conv7_head_dgrad_lp16_supported": (_i, [_DP]),
    "mmh_conv7_head_dgrad_lp16_ws_bytes": (_sz, [_DP]),
    "mmh_conv7_head_dgrad_lp16": (_i, [_DP, _vp, _vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "mmh_conv7_n4_lp16_supported": (_i, [_DP, _i]),
    "mmh_conv7_n4_lp16_ws_bytes": (_sz, [_DP, _i]),
    "mmh_conv7_n4_lp16": (_i, [_DP, _i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "mmh_conv3x3_lp16_supported": (_i, [_DP]),
    "mmh_conv3x3_lp16_fold_supported": (_i, [_DP]),
    "mmh_wgrad3x3_lp16_ws_bytes": (_sz, [_DP]),
    "mmh_wgrad3x3_lp16": (_i, [_DP, _vp, _vp, _vp, _vp, _sz, _i, _vp, _vp]),
    "mmh_conv3x3_lp16": (_i, [_DP, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "mmh_conv3x3_lp16_stats_chunks": (_i, [_DP]),
    "mmh_conv3x3_lp16_dgrad_nbr_chunks": (_i, [_DP, _i]),
    "mmh_conv3x3_lp16_dgrad_nbr": (_i, [_DP, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp, _vp]),
    "mmh_norm_bwd_sums_final": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mmh_conv3x3_lp16_dgrad_add_supported": (_i, [_DP]),
    "mmh_conv3x3_lp16_dgrad_add": (_i, [_DP, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmh_conv3x3_lp16_fprop_stats": (_i, [_DP, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mmh_colsum_ws_bytes": (_sz, [_i64, _i]),
    "mmh_colsum": (_i, [_vp, _i64, _i, _i, _vp, _vp, _sz, _i, _i, _vp]),
    "mmh_norm_stats_ws_bytes": (_sz, [_i, _i64, _i]),
    "mmh_norm_stats": (_i, [_vp, _i, _i64, _i, _i, _vp, _vp, _vp, _sz, _i, _vp]),
    "mmh_norm_finalize": (_i, [_vp, _vp, _d, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp]),
    "mmh_syncbn_merge_finalize": (_i, [_vp, _i, _i64, _i, _d, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp]),
    "mmh_scale_shift_act": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _u64, _vp, _vp, _i, _i, _vp]),
    "mmh_scale_shift_act_twin": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _u64, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "mmh_norm_bwd_ws_bytes": (_sz, [_i, _i64, _i]),
    "mmh_norm_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _vp, _vp, _vp, _sz, _i, _i, _vp]),
    "mmh_norm_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i64, _i, _i, _f, _vp, _i, _i, _i, _vp]),
    "mmh_norm_bwd_fused_supported": (_i, [_i, _i64, _i, _i, _i, _i]),
    "mmh_norm_bwd_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i64, _i, _i, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mmh_dropout_bits": (_i, [_i64, _f, _u64, _vp, _vp, _vp]),
    "mmh_dropout_bits_rows": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "mmh_dropout_bits_both": (_i, [_i64, _i, _i, _f, _u64, _vp, _vp, _vp, _vp]),
    "mmh_norm_bwd_reduce_rc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "mmh_norm_bwd_apply_rc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i64, _i, _i, _f, _vp, _vp]),
    "mmh_wino_input_normact": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _f, _vp, _vp]),
    "mmh_wino_input_dy_normbwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp,
                                       _vp, _i, _i, _f, _vp]),
    "mmh_act_bwd": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "mmh_act_bwd_lp16": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "mmh_act_bwd_lp16_io": (_i, [_vp, _i, _vp, _i, _i64, _i, _i, _vp, _vp]),
    "mmh_patblock_gate_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "mmh_patblock_gate_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "mmh_patblock_gate_norm_supported": (_i, [_i, _i64, _i]),
    "mmh_patblock_gate_norm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _vp]),
    "mmh_patblock_gate_norm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                        _i, _i64, _i, _i, _i, _i, _vp]),
    "mmh_reduce_ws_bytes": (_sz, [_i64]),
    "mmh_bce_logits_fwd": (_i, [_vp, _i64, _f, _f, _d, _vp, _vp, _sz, _vp]),
    "mmh_bce_logits_bwd": (_i, [_vp, _i64, _f, _f, _d, _vp, _vp, _vp]),
    "mmh_l1_fwd": (_i, [_vp, _vp, _i64, _f, _d, _vp, _vp, _sz, _vp]),
    "mmh_l1_bwd": (_i, [_vp, _vp, _i64, _f, _d, _vp, _vp, _vp]),
    "mmh_l1_fwd_lp16": (_i, [_vp, _vp, _i64, _f, _d, _i, _vp, _vp, _sz, _vp]),
    "mmh_l1_relu_bwd_lp16": (_i, [_vp, _vp, _i64, _f, _d, _vp, _i, _vp, _vp]),
    "mmh_mse_fwd": (_i, [_vp, _vp, _i64, _f, _d, _vp, _vp, _sz, _vp]),
    "mmh_maxpool2x2_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mmh_maxpool2x2_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "mmh_mse_bwd": (_i, [_vp, _vp, _i64, _f, _d, _vp, _vp, _vp]),
    "mmh_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f, _vp, _vp, _vp]),
    "mmh_grad_nonfinite": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "mmh_loss_scale_update": (_i, [_vp, _vp, _f, _f, _i, _f, _f, _vp]),
    "mmh_pack_nhwc": (_i, [C.POINTER(PlaneSrc), _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mmh_pack_nhwc_lp16": (_i, [C.POINTER(PlaneSrc), _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mmh_pose_heatmaps": (_i, [_vp, _i, _i, _i, _d, _vp, _vp]),
    "mmh_map_to_cord": (_i, [_vp, _i, _i, _i, _f, _vp, _vp]),
    "mmh_decode_inputs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "mmh_rccl_bind": (_i, [C.c_char_p]),
    "mmh_rccl_comm_ranks": (_i, [_vp]),
    "mmh_allreduce_bucket": (_i, [_vp, _vp, _i64, _i, _vp]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mmhand_amd/csrc`.  There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # MMH_OPTIONS="key=value,key=value": kernel-selection switches of mmh_set_option for a whole run (A/B on a node this code
    # cannot be edited on, e.g. "lp16_persist=0" under data parallelism when the collectives' kernels hold CUs); an unknown key
    # or a malformed entry stops the run
    for item in filter(None, (x.strip() for x in os.environ.get("MMH_OPTIONS", "").split(","))):
        key, sep, val = item.partition("=")
        try:
            ival = int(val)
        except ValueError:
            ival = None
        if not sep or ival is None:
            raise RuntimeError(f"MMH_OPTIONS: '{item}' is not key=integer")
        check(lib.mmh_set_option(key.strip().encode(), ival), f"MMH_OPTIONS {item}")
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mmh_last_error()
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")


def call(name, *args):
    """Invoke a status-returning entry point and raise on error."""
    rc = getattr(_lib or load(), name)(*args)
    if rc != 0:
        check(rc, name)
