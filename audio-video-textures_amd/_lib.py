"""ctypes loader of the gfx950 C-ABI library (include/avt.h).

There is no CPU fallback: if libavt_hip.so is missing or a call fails, the
product path raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C audio-video-textures_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AVT_HIP_LIB selects another build of the same ABI (the diagnostic libavt_hip_stamp.so of `make stamp`); default = the product
LIB_PATH = os.environ.get("AVT_HIP_LIB") or os.path.join(_HERE, "libavt_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "avt.h")

_i32p, _i64p, _f32p, _vp = (C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_void_p)

# name -> argtypes; mirrors include/avt.h one to one (tests/test_abi.py checks the header against it)
SIGNATURES = {
    "avt_abi_version": [],
    "avt_device_check": [C.c_char_p, C.c_size_t],
    "avt_clip_sample_table": [C.c_int, _vp, _vp],
    "avt_clip_pack_plan": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "avt_clip_pack_u8": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float,
                         C.c_int, _vp, _vp, C.c_int, _vp],
    "avt_l2norm_rows": [_vp, C.c_int, _vp, C.c_int, C.c_int64, C.c_float, _vp, _vp, _vp, _vp],
    "avt_sim_gemm_nt": [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int, C.c_float, C.c_int, _vp, C.c_int64, _vp],
    "avt_gemm_nt_x3_f32out": [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, _vp, C.c_int, _vp],
    "avt_row_transition": [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_float,
                           C.c_float, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "avt_row_topk": [_vp, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_int, _vp, _vp, _vp],
    "avt_softmax_ce_fwd": [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp],
    "avt_softmax_ce_bwd": [_vp, _vp, C.c_int64, C.c_int64, C.c_float, _vp, _vp],
    "avt_clip_pack_u8_ndhwc4": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                C.c_int, _vp, _vp, _vp],
    "avt_maxpool_hw3s2_ndhwc_bf16": [_vp, _vp] + [C.c_int] * 7 + [_vp],
    "avt_mean_positions_bf16": [_vp] + [C.c_int] * 4 + [_vp, C.c_int, _vp],
    "avt_maxpool_hw2s2_ndhwc_bf16": [_vp, _vp] + [C.c_int] * 6 + [_vp],
    "avt_stem_conv_supported": [C.c_int] * 3,
    "avt_stem_conv_bf16": [_vp, _vp, _vp, _vp] + [C.c_int] * 9 + [_vp],
    "avt_stem_conv_pool_bf16": [_vp, _vp, _vp, _vp] + [C.c_int] * 10 + [_vp],
    "avt_bottleneck_fused_supported": [C.c_int] * 2,
    "avt_bottleneck_fused_bf16": [_vp] * 8 + [C.c_int] * 6 + [_vp],
    "avt_bottleneck_first_supported": [C.c_int] * 3,
    "avt_bottleneck_first_bf16": [_vp] * 9 + [C.c_int] * 7 + [_vp],
    "avt_pairwise_l2_f32": [_vp, C.c_int, C.c_int64, _vp, _vp],
    "avt_diag_filter_f32": [_vp, C.c_int, _vp, C.c_int, _vp, _vp],
    "avt_q_learning_supported": [C.c_int],
    "avt_q_learning_f32": [_vp, C.c_int, C.c_float, C.c_float, C.c_int, _vp, _vp, _vp],
    "avt_conv33_c64_supported": [C.c_int] * 3,
    "avt_conv33_c64_bf16": [_vp] * 4 + [C.c_int] * 6 + [_vp],
    "avt_pw_chain_supported": [C.c_int] * 5,
    "avt_pw_chain_bf16": [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int,
                          _vp, _vp, _vp, C.c_int, C.c_int, C.c_int64, _vp],
    "avt_logmel_f64": [_vp, C.c_int, C.c_int64, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_double, _vp, _vp],
    "avt_logmel_examples_f32": [_vp, C.c_int64, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "avt_infonce_fwd": [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, _vp, _vp, _vp, _vp],
    "avt_infonce_bwd": [_vp] * 6 + [C.c_int64, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp],
    "avt_conv3d_ktab": [C.c_int] * 7 + [_vp, C.c_int],
    "avt_conv3d_igemm_bf16": [_vp] * 6 + [C.c_int] * 22 + [_vp],
    "avt_conv3d_igemm_rows_bf16": [_vp] * 6 + [C.c_int] * 25 + [_vp],
    "avt_bn_train_fwd": [_vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_float, C.c_float, C.c_int, C.c_int, _vp, C.c_int64, _vp, _vp, _vp,
                         _vp, _vp, _vp, C.c_int64, _vp],
    "avt_bn_train_bwd": [_vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp,
                         C.c_int64, _vp],
    "avt_bn_train_ws_bytes": [C.c_int64, C.c_int, C.c_int],
    "avt_bn_train_ws_bytes_pre": [C.c_int, C.c_int, C.c_int],
    "avt_bn_train_fwd_pre": [_vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_float, C.c_float, C.c_int, C.c_int, _vp, C.c_int64, _vp, _vp,
                             _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp],
    "avt_conv3d_igemm_x3_f32_stat_rows": [C.c_int, C.c_int, C.c_int64, C.c_int],
    "avt_conv3d_igemm_x3_f32_bwdstats_rows": [C.c_int, C.c_int, C.c_int64, C.c_int],
    "avt_conv3d_igemm_x3_f32_bwdstats": [_vp] * 7 + [C.c_int] * 16 + [_vp] * 6 + [C.c_int, _vp, C.c_int, C.c_int, _vp],
    "avt_bn_train_bwd_pre": [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, C.c_int, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp],
    "avt_pw_x3_f32_stat_rows": [C.c_int, C.c_int, C.c_int64, C.c_int],
    "avt_pw_x3_f32_bwdstats_rows": [C.c_int, C.c_int, C.c_int64, C.c_int],
    "avt_pw_x3_f32_bwdstats": [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int64, C.c_int] + [_vp] * 6 +
                              [C.c_int, _vp, C.c_int, _vp],
    "avt_pw_x3_f32_stats": [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int64, C.c_int, _vp, C.c_int, _vp],
    "avt_conv3d_igemm_x3_f32_stats": [_vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 18 + [_vp, C.c_int, C.c_int, _vp],
    "avt_conv3d_igemm_x3_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 19 + [_vp],
    "avt_conv3d_igemm_x3_f32_ex": [_vp] * 6 + [C.c_int] * 21 + [_vp],
    "avt_conv3d_wgrad_x3_f32": [_vp, _vp, _vp] + [C.c_int] * 17 + [_vp],
    "avt_wgrad_x3_set_xl": [C.c_int],
    "avt_conv_x3_set_small_tile": [C.c_int],
    "avt_conv3d_wgrad_x3_sub_f32": [_vp, _vp, _vp] + [C.c_int] * 22 + [_vp],
    "avt_interp_pack_pair_u8": [_vp, _vp, C.c_int, C.c_int, _f32p, _vp, _vp, _vp, C.c_int, _vp],
    "avt_avgpool2_x3": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp],
    "avt_upsample2_bilinear_x3": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp],
    "avt_interp_mid_input": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp],
    "avt_interp_final_u8": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32p, _vp, C.c_int, _vp],
    "avt_negative_sample_mt19937": [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "avt_clip_pack_gather_u8": [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                C.c_int, _vp, _vp, C.c_int, _vp],
    "avt_stem_conv_x3": [_vp] * 8 + [C.c_int] * 11 + [_vp, C.c_int, _vp],
    "avt_lateral_x3_supported": [C.c_int] * 3,
    "avt_lateral_x3": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp] + [C.c_int] * 10 + [_vp],
    "avt_stem_conv_x3_merged": [_vp] * 8 + [C.c_int] * 10 + [_vp, _vp, C.c_int, C.c_int, _vp],
    "avt_clip_planes_f32": [_vp] + [C.c_int] * 4 + [C.c_int64] * 5 + [_vp, _vp, C.c_int, _vp],
    "avt_stem_conv_x3_f32": [_vp] * 6 + [C.c_int] * 11 + [_vp],
    "avt_weight_planes_f32": [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp],
    "avt_weight_planes_t_f32": [_vp, C.c_int, C.c_int, C.c_int, _i32p, C.c_int, _vp, _vp, _vp],
    "avt_weight_planes_gather_f32": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp],
    "avt_weight_planes_job_bytes": [],
    "avt_weight_planes_multi": [_vp, _vp, C.c_int, _vp],
    "avt_maxpool_train_fwd": [_vp, _vp, _vp] + [C.c_int] * 4 + [C.c_int64, _vp],
    "avt_maxpool_train_bwd": [_vp, _vp, _vp] + [C.c_int] * 4 + [C.c_int64, _vp],
    "avt_stem_wgrad_x3_supported": [C.c_int] * 4,
    "avt_stem_wgrad_x3": [_vp] * 4 + [C.c_int] * 7 + [_vp],
    "avt_pw_x3_supported": [C.c_int] * 2,
    "avt_pw_x3": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int64,
                  C.c_int, C.c_int, _vp],
    "avt_pw_x3_f32_supported": [C.c_int] * 2,
    "avt_pw_x3_f32": [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int64, C.c_int, _vp],
    "avt_pw_chain_x3_supported": [C.c_int] * 3,
    "avt_pw_chain_x3": [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int,
                        _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int64, C.c_int, _vp],
    "avt_conv3d_igemm_x3": [_vp] * 10 + [C.c_int] * 26 + [_vp, _vp],
    "avt_conv3d_igemm_x3_wblk": [_vp] * 10 + [C.c_int] * 26 + [_vp, _vp],
    "avt_clip_pack_u8_ndhwc4_x3": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                   C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp],
    "avt_maxpool_hw3s2_ndhwc_x3": [_vp] * 4 + [C.c_int] * 8 + [_vp, _vp],
    "avt_maxpool_hw2s2_ndhwc_x3": [_vp] * 4 + [C.c_int] * 7 + [_vp],
    "avt_mean_positions_x3": [_vp, _vp] + [C.c_int] * 4 + [_vp, C.c_int, C.c_int, _vp],
    "avt_conv3d_igemm_x3_xl_picked": [C.c_int] * 3,
    "avt_conv33_x3_supported": [C.c_int] * 2,
    "avt_conv33_x3": [_vp] * 6 + [C.c_int] * 8 + [_vp],
    "avt_bneck_x3_supported": [C.c_int] * 3,
    "avt_bneck_x3": [_vp] * 6 + [C.c_int] * 8 + [_vp],
    "avt_res2_x3_supported": [C.c_int] * 3,
    "avt_res2_x3_wfrag_bytes": [],
    "avt_res2_x3": [_vp] * 6 + [C.c_int] * 7 + [_vp],
    "avt_conv3d_igemm_wfrag_supported": [C.c_int] * 5,
    "avt_conv3d_igemm_wfrag_bf16": [_vp] * 6 + [C.c_int] * 25 + [_vp, C.c_int, _vp],
}


class AvtError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded library; raises (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AvtError(
                "HIP extension not built: %s is missing. Run `make -C %s` (needs hipcc); "
                "there is no CPU fallback for the hot path." % (LIB_PATH, os.path.join(_HERE, "csrc")))
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int64 if name in _RETURNS_I64 else C.c_int
        handle.avt_last_error.argtypes = []
        handle.avt_last_error.restype = C.c_char_p
        if handle.avt_abi_version() != ABI_VERSION:
            raise AvtError("libavt_hip.so ABI version %d, expected %d: rebuild with `make -C %s`"
                           % (handle.avt_abi_version(), ABI_VERSION, os.path.join(_HERE, "csrc")))
        if os.environ.get("AVT_SMALL_TILE", "") == "0":  # (A/Bs: the training convolutions' 64-row tile at small batches off)
            handle.avt_conv_x3_set_small_tile(0)
        if os.environ.get("AVT_WGRAD_ROUNDS", "") == "0":  # (A/Bs: the weight gradient's round-aware split over positions off)
            handle.avt_wgrad_x3_set_xl(7)
        _lib = handle
    return _lib


ABI_VERSION = 8  # include/avt.h AVT_ABI_VERSION
_RETURNS_I64 = {"avt_bn_train_ws_bytes", "avt_bn_train_ws_bytes_pre"}  # sizes; every other entry returns an AVT_* status


def check(status, what):
    if status != 0:
        msg = lib().avt_last_error()
        raise AvtError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))
