"""ctypes binding of libpram_hip.so (the C ABI declared in include/pram_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os as _os
# PRAM_HIP_LIB: an alternative build of the same library (A/B timing of build options under profiles/); never set in production
_LIB_PATH = Path(_os.environ["PRAM_HIP_LIB"]) if _os.environ.get("PRAM_HIP_LIB") else Path(__file__).resolve().parent / "csrc" / "libpram_hip.so"
_lib = None

P = C.c_void_p
I = C.c_int
F = C.c_float
LL = C.c_longlong
SZ = C.c_size_t

_SIGS = {
    "pram_hip_version": (I, []),
    "pram_last_error": (C.c_char_p, []),
    "pram_set_status_word": (I, [P]),
    "pram_read_status_word": (I, [P, I, P]),
    "pram_x3_set_act_scale": (F, [F]),
    "pram_linear_f32": (I, [P, I, I, P, I, I, P, P, P, I, P, I, I, I, F, I, P, P, I, P]),
    "pram_linear_f16_f32": (I, [P, I, I, P, I, I, P, P, P, I, P, I, I, I, F, I, P, P, I, P]),
    "pram_linear_f16_ragged_f32": (I, [P, I, I, P, I, I, P, P, P, I, P, I, I, I, F, I, P, P, I, P, I, P]),
    "pram_linear_f16_h16": (I, [P, I, I, P, I, I, P, P, P, I, P, I, P, I, I, I, F, I, P, P, I, P]),
    "pram_linear_x3_f32": (I, [P, I, I, P, I, I, P, P, F, P, P, I, P, I, P, P, I, I, I, F, I, P, P, I, P]),
    "pram_linear_ragged_f32": (I, [P, I, I, P, I, I, P, P, P, I, P, I, I, I, F, I, P, P, I, P, I, P]),
    "pram_linear_x3_ragged_f32": (I, [P, I, I, P, I, I, P, P, F, P, P, I, P, I, P, P, I, I, I, F, I, P, P, I, P, I, P]),
    "pram_debug_gemm_phases": (I, [P, I]),
    "pram_linear_x3_ssq_parts": (I, [I, I, I]),
    "pram_linear_x3_ssq_f32": (I, [P, I, I, P, I, I, P, P, F, P, P, I, P, I, I, P, I, P]),
    "pram_linear_x3_lngelu_f32": (I, [P, I, I, P, P, F, P, P, I, P, I, I, I, P, I, P, P, F, P, I, P]),
    "pram_linear_x3_qkv_f32": (I, [P, I, I, P, P, F, P, P, P, I, P, P, I, I, I, I, I, I, P, P, I, P, P]),
    "pram_layernorm_gelu_ragged_f32": (I, [P, I, P, I, P, P, I, I, F, P, I, P]),
    "pram_linear_x3p_f32": (I, [P, P, I, I, P, P, I, I, P, P, F, P, P, I, P, I, P, P, I, I, I, F, I, P, P, I, P]),
    "pram_attention_x3_f32": (I, [P, P, I, P, P, I, P, P, P, I, P, P, P, I, I, I, I, F, I, P, SZ, P]),
    "pram_attention_x3_workspace_bytes": (SZ, [I, I, I, I]),
    "pram_attention_x3_mfma_per_tile": (I, [I]),
    "pram_attention_x3_is_split": (I, [I, I, I, I]),
    "pram_attention_x3_set_split_target": (I, [I]),
    "pram_attention_x3_set_chunk_keys": (I, [I]),
    "pram_attention_x3_set_p_split": (I, [I]),
    "pram_attention_x3_colmean_f32": (I, [P, P, I, P, P, I, P, P, P, P, I, I, I, I, F, I, P]),
    "pram_attention_h16t_f32": (I, [P, I, P, I, P, P, I, P, P, P, I, I, I, I, F, I, P]),
    "pram_attention_x3_vt": (I, [P, P, I, P, P, P, I, I, I, P]),
    "pram_conv2d_nhwc_x3_l2norm_f32": (I, [P, I, I, I, I, P, P, F, P, P, P, P, P, I, I, I, I, P]),
    "pram_conv2d_nhwc_x3_planes": (I, [P, I, I, I, I, P, P, F, P, P, P, P, P, P, I, I, I, I, P]),
    "pram_conv3x3_grouped_planes_x3_f32": (I, [P, P, I, I, I, I, P, P, F, P, P, P, I, I, P]),
    "pram_sfd2_conv1_x3_f32": (I, [P, I, I, I, P, P, F, P, P, P, P, P, F, P, P, P, P, I, P]),
    "pram_conv2d_nhwc_x3_f32": (I, [P, I, I, I, I, P, P, F, P, P, P, P, P, I, I, I, I, P]),
    "pram_bgemm_nt_x3p_f32": (I, [P, P, I, LL, P, P, I, LL, P, I, LL, I, I, I, I, F, P]),
    "pram_bgemm_nt_f32": (I, [P, I, LL, P, I, LL, P, I, LL, I, I, I, I, F, P]),
    "pram_layernorm_gelu_f32": (I, [P, I, P, I, P, P, I, I, F, P]),
    "pram_fourier_encoding_f32": (I, [P, P, F, F, F, P, P, I, P]),
    "pram_attention_f32": (I, [P, I, P, I, P, I, P, I, P, P, P, I, I, I, I, F, P, SZ, P]),
    "pram_attention_workspace_bytes": (SZ, [I, I, I, I]),
    "pram_attention_h16_f32": (I, [P, I, P, I, P, I, P, I, P, P, P, I, I, I, I, F, I, P]),
    "pram_attention_f16_f32": (I, [P, I, P, I, P, I, P, I, P, P, P, I, I, I, I, F, P]),
    "pram_attention_colmean_f32": (I, [P, I, P, I, P, P, P, P, I, I, I, I, F, P]),
    "pram_attention_cross_f32": (I, [P, I, P, I, P, I, P, P, I, I, I, F, P, SZ, P]),
    "pram_attention_cross_f16_f32": (I, [P, I, P, I, P, I, P, P, I, I, I, F, P]),
    "pram_attention_cross_colmean_f32": (I, [P, I, P, P, P, I, I, I, F, P]),
    "pram_sinkhorn_workspace_bytes": (SZ, [I, I, I]),
    "pram_sinkhorn_match_f32": (I, [P, I, P, P, P, I, F, P, I, P, P, P, P, I, I, I, P, P]),
    "pram_dual_softmax_match_f32": (I, [P, I, P, P, P, F, P, I, P, P, P, P, I, I, I, P, P]),
    "pram_adagml_prune_f32": (I, [P, F, I, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "pram_adagml_layer_state": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "pram_adagml_scores4_f32": (I, [P, P, P, LL, P]),
    "pram_adagml_prune_ld_f32": (I, [P, I, F, I, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "pram_adagml_scatter_f32": (I, [P, P, P, P, P, I, I, I, P, P, P]),
    "pram_conv2d_nhwc_f32": (I, [P, I, I, I, I, P, P, P, P, P, P, I, I, I, I, P]),
    "pram_conv2d_nhwc_f16_f32": (I, [P, I, I, I, I, P, P, P, P, P, P, I, I, I, I, P]),
    "pram_conv3x3_grouped_nhwc_f32": (I, [P, I, I, I, I, P, P, P, P, I, I, P]),
    "pram_image_to_nhwc4_f32": (I, [P, P, I, I, I, P]),
    "pram_nhwc_to_nchw_f32": (I, [P, P, I, I, I, I, P]),
    "pram_score_map_f32": (I, [P, P, I, I, I, P]),
    "pram_simple_nms_workspace_bytes": (SZ, [I, I, I]),
    "pram_simple_nms_f32": (I, [P, P, I, I, I, I, P, P]),
    "pram_select_keypoints_workspace_bytes": (SZ, [I, I, I, I]),
    "pram_select_keypoints_f32": (I, [P, I, I, I, F, I, I, I, I, P, P, P, P, P]),
    "pram_sample_nhwc_f32": (I, [P, I, I, I, I, P, P, I, I, I, P, P]),
    "pram_l2norm_rows_f32": (I, [P, I, I, P]),
    "pram_resize_bilinear_f32": (I, [P, P, I, I, I, I, I, P]),
    "pram_seg_epilogue_f32": (I, [P, P, I, I, I, F, P, P, P, P, P]),
    "pram_row_sort_desc_f32": (I, [P, I, I, I, P, P, P]),
    "pram_row_top2_f32": (I, [P, I, LL, P, P, I, I, I, I, P, P, P, P]),
    "pram_proj_dist_top2_f32": (I, [P, I, P, P, I, I, F, P, P, P, P]),
    "pram_proj_dist_top2_f64uv": (I, [P, I, P, P, I, I, I, C.c_double, P, P, P, P]),
    "pram_project_points_f64": (I, [P, P, P, I, C.c_double, C.c_double, P, P, P, P, P, P]),
    "pram_seg_vote": (I, [P, P, I, I, I, P, P, P, P, P, P, P]),
    "pram_linear_f16_qkv_h16": (I, [P, I, I, P, P, P, I, P, I, I, I, I, I, I, P, P, I, P, P]),
    "pram_attention_h16t_h16": (I, [P, I, P, I, P, P, I, P, P, P, I, I, I, I, F, I, P]),
    "pram_linear_f16_ssq_h16": (I, [P, I, I, P, I, I, P, P, P, I, P, I, I, P]),
    "pram_linear_f16_lngelu_f32": (I, [P, I, I, P, P, P, I, P, I, I, I, P, I, P, P, F, P]),
    "pram_pack_record_f32": (I, [P, P, P, P, P, I, I, I, P, P]),
    "pram_fill_u32": (I, [P, C.c_uint, SZ, P]),
    "pram_stage_frames_u8": (I, [P, P, P, I, I, I, P]),
    "pram_score_lookup_f32": (I, [P, LL, I, I, P, P, I, I, P, P]),
}


class PramHipError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load libpram_hip.so; raises PramHipError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise PramHipError(
            f"{_LIB_PATH} not found: build it with `python -m pram_amd.build` "
            "(pram_amd has no CPU / eager fallback)")
    # PyTorch-ROCm ships its own libamdhip64 and owns the device context the kernels launch into: it has to be the
    # first HIP runtime in the process.  (Loading this library first binds it to /opt/rocm's copy, and every launch
    # then fails with "no ROCm-capable device is detected" once torch has initialised its own.)
    import torch  # noqa: F401
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGS)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pram_last_error()
        raise PramHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
