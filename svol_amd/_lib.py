"""ctypes binding of libsvol_hip.so (include/svol_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails the
product raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is
never imported from here.)
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsvol_hip.so')

_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_int = ctypes.c_int
_f32 = ctypes.c_float
_u64 = ctypes.c_uint64

# name -> argtypes, exactly the declarations of include/svol_hip.h
SIGNATURES = {
    'svol_block_trace_dump': [ctypes.c_char_p, _i64],
    'svol_lsap_solve': [_p, _i64, _i64, _p, _p],
    'svol_heads_fwd': [_p] * 13 + [_i64, _i64, _p],
    'svol_heads_bwd': [_p] * 17 + [_i64, _i64, _p],
    'svol_set_loss_bwd': [_p] * 6 + [_i64, _i64, _p],
    'svol_weighted_total': [_p, _p, _i64, _p, _p],
    'svol_weighted_total_bwd': [_p, _p, _i64, _p, _p],
    'svol_cast': [_p, _int, _p, _int, _i64, _p],
    'svol_cast_transpose': [_p, _p, _p, _int, _i64, _i64, _p],
    'svol_cast_split': [_p, _i64, _p, _i64, _i64, _p],
    'svol_gemm_nt_split': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _p],
    'svol_gemm_nt': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _int, _p, _i64, _p, _i64, _int, _i64, _i64, _i64,
                     _int, _p],
    'svol_gemm_nt_dgelu': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _int, _p],
    'svol_gemm_nt_dact': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _int, _p, _i64, _i64, _i64, _int, _p],
    'svol_gemm_tn': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _int, _p],
    'svol_gemm_tn_grouped': [_p, _i32, _int, _p],
    'svol_colsum': [_p, _i64, _p, _i64, _i64, _int, _p],
    'svol_act_bwd': [_p, _p, _p, _int, _i64, _int, _p],
    'svol_layernorm_fwd': [_p, _int, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _f32, _u64, _p, _int, _p],
    'svol_layernorm_bwd': [_p, _p, _p, _p, _int, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f32, _u64, _p, _int, _p],
    'svol_cast_transpose_multi': [_p, _int, _i64, _int, _p],
    'svol_postprocess': [_p, _p, _p, _i64, _i64, _i64, _p],
    'svol_eval_max_iou': [_p, _p, _p, _p, _p, _p, _i64, _int, _p],
    'svol_eval_ap': [_p, _p, _p, _p, _p, _p, _p, _p, _int, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    'svol_patchify': [_p, _p, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_vit_embed': [_p, _p, _p, _p, _p, _i64, _i64, _i64, _int, _p],
    'svol_adamw_flat': [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _p],
    'svol_adamw_flat_zero': [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _i64, _f32, _p],
    'svol_grad_finite': [_p, _i64, _p, _p],
    'svol_adamw_flat_scaled': [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _p, _p],
    'svol_loss_scaler_update': [_p, _f32, _f32, _i64, _f32, _f32, _p],
    'svol_conv_nhwc': [_p, _p, _i64, _p, _p, _int, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_im2col': [_p, _i64, _i64, _i64, _i64, _int, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_maxpool_nhwc': [_p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_avgpool_nhwc': [_p, _p, _i64, _i64, _i64, _int, _p],
    'svol_bn_colstats': [_p, _p, _f32, _p, _p, _i64, _i64, _int, _p],
    'svol_bn_finalize': [_p, _p, _p, _p, _p, _p, _p, _f32, _f32, _i64, _i64, _p, _p, _p, _p, _p],
    'svol_conv_weight_pack': [_p, _p, _i64, _i64, _i64, _i64, _i64, _int, _int, _p],
    'svol_conv_weight_unpack_add': [_p, _p, _i64, _i64, _i64, _i64, _i64, _p],
    'svol_conv_wgrad_nhwc': [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_bn_apply': [_p, _p, _p, _p, _int, _p, _i64, _i64, _int, _p],
    'svol_bn_bwd_reduce': [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _int, _p],
    'svol_bn_bwd_apply': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _int, _p],
    'svol_col2im_nhwc': [_p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_maxpool_idx_nhwc': [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_maxpool_bwd_nhwc': [_p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_attn_weights_mean': [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _int, _p],
    'svol_attn_small_fwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _f32, _int, _p],
    'svol_posenc_sine': [_p, _p, _i64, _i64, _i64, _int, _p],
    'svol_attn_ws_bytes': [_i64, _i64, _i64, _i64, _i64],
    'svol_attn_fwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _p, _i64,
                      _int, _p],
    'svol_attn_bwd': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64,
                      _i64, _i64, _i64, _i64, _i64, _f32, _f32, _p, _i64, _int, _p],
    'svol_attn_bwd_ex': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64,
                         _i64, _i64, _i64, _i64, _i64, _f32, _f32, _p, _i64, _int, _int, _p, _p],
    'svol_attn_bwd_sp_image_bytes': [_i64, _i64, _i64, _i64, _i64, _i64, _int],
    'svol_attn_bwd_zero_ws': [_p, _i64, _i64, _i64, _i64, _i64, _i64, _int, _p],
    'svol_attn_fwd_dropout': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i64, _i64, _f32, _f32, _p, _i64,
                              _f32, _u64, _int, _p],
    'svol_attn_bwd_dropout': [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64,
                              _i64, _i64, _i64, _i64, _i64, _f32, _f32, _p, _i64, _f32, _u64, _int, _p],
    'svol_dropout': [_p, _p, _i64, _i64, _f32, _u64, _int, _p],
    'svol_dropout_add': [_p, _p, _p, _i64, _i64, _f32, _u64, _p],
    'svol_gate_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _int, _p],
    'svol_gate_bwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _int,
                      _p],
    'svol_clock_probe': [_p, _p, _int, _p, _i64, _i64, _p, _p],
    'svol_gate_fwd_scored': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _int, _p],
    'svol_layernorm_gate_scores_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _int, _p],
    'svol_gate_vectors_fwd': [_p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    'svol_gate_vectors_bwd': [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p],
    'svol_gate_vectors_fwd_multi': [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    'svol_gate_vectors_bwd_multi': [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _p],
    'svol_match_cost': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _f32, _f32, _f32, _p, _p],
    'svol_lsap_batched': [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p],
    'svol_set_loss': [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _f32, _p, _i32, _p, _p, _i32, _p],
    # composite block programs: (dims int64[], slots void*[], [phase,] stream)
    'svol_video_half_fwd': [_p, _p, _p],
    'svol_video_half_bwd': [_p, _p, _int, _p],
    'svol_video_half_wgrad': [_p, _p, _p],
    'svol_video_half_wgrad_part': [_p, _p, _int, _p],
    'svol_query_self_fwd': [_p, _p, _p],
    'svol_query_self_bwd': [_p, _p, _p],
    'svol_query_self_wgrad': [_p, _p, _p],
    'svol_query_cross_fwd': [_p, _p, _p],
    'svol_query_cross_bwd': [_p, _p, _p],
    'svol_query_cross_wgrad': [_p, _p, _p],
}



class TnProblem(ctypes.Structure):
    """svol_tn_problem (include/svol_hip.h): one weight-gradient GEMM of a grouped launch."""
    _fields_ = [('A', _p), ('lda', _i64), ('B', _p), ('ldb', _i64), ('C', _p), ('ldc', _i64), ('colsum', _p),
                ('Mc', _i64), ('N', _i64), ('K', _i64)]


_LIB = None


class SvolHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library; raise loudly when absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SvolHipError(
                f'{LIB_PATH} not found: the SVOL hot path has no CPU fallback. '
                'Build it with `python -m svol_amd.build` (hipcc, gfx950).')
        L = ctypes.CDLL(LIB_PATH)
        L.svol_abi_version.restype = _int
        L.svol_abi_version.argtypes = []
        L.svol_strerror.restype = ctypes.c_char_p
        L.svol_strerror.argtypes = [_int]
        for name, at in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the symbol is not exported
            f.restype = _int
            f.argtypes = at
        L.svol_attn_ws_bytes.restype = _i64
        L.svol_attn_bwd_sp_image_bytes.restype = _i64
        L.svol_block_slot_names.restype = ctypes.c_char_p
        L.svol_block_slot_names.argtypes = [_int]
        _LIB = L
    return _LIB


def check(rc: int, name: str):
    if rc != 0:
        msg = lib().svol_strerror(rc).decode()
        raise SvolHipError(f'{name} failed: {msg} (code {rc})')
