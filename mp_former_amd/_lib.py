"""ctypes binding of libmpformer_hip.so — the C-ABI boundary (include/mpformer_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C mp_former_amd/csrc`` and is
loaded lazily on first use.  A missing library is a hard error: there is no fallback path.
"""
import ctypes

import torch
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpformer_hip.so")
# development only: A/B a second build of the library in one process run (tools/ab_lib.sh)
LIB_PATH = os.environ.get("MPF_LIB_PATH", LIB_PATH)
ABI_VERSION = 1

MPF_F32, MPF_F64, MPF_BF16, MPF_U8, MPF_BITS = 0, 1, 2, 3, 4

_lib = None

_c_int = ctypes.c_int
_c_vp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol declared in include/mpformer_hip.h
SIGNATURES = {
    "mpf_abi_version": (_c_int, []),
    "mpf_last_error": (ctypes.c_char_p, []),
    "mpf_last_kernel": (ctypes.c_char_p, []),
    "mpf_set_option": (_c_int, [ctypes.c_char_p, _c_int]),
    "mpf_match_cost_fused_workspace_bytes": (ctypes.c_size_t, [_c_int] * 5),
    "mpf_match_cost_fused": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int,
                                      _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_vp,
                                      ctypes.c_size_t, _c_vp]),
    "mpf_pair_planes_forward": (_c_int, [_c_vp] * 6 + [ctypes.c_int64, _c_vp, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_pair_planes_backward_workspace_bytes": (ctypes.c_size_t, [_c_int] * 4),
    "mpf_pair_planes_backward": (_c_int, [_c_vp] * 7 + [ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_int, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_msda_dev_workspace_bytes": (ctypes.c_size_t, [_c_int] * 7),
    "mpf_msda_forward_dev": (_c_int, [_c_vp] * 6 + [_c_int] * 8 + [_c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_msda_backward_dev": (_c_int, [_c_vp] * 9 + [_c_int] * 8 + [_c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_msda_dev_geometry": (_c_int, [_c_vp, ctypes.POINTER(_c_int), _c_int]),
    "mpf_msda_stats": (_c_int, [ctypes.POINTER(ctypes.c_ulonglong), _c_int, _c_int]),
    "mpf_profile_enable": (_c_int, [_c_int]),
    "mpf_profile_get_flops": (_c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]),
    "mpf_profile_get": (_c_int, [ctypes.c_char_p, ctypes.POINTER(_c_int), ctypes.POINTER(ctypes.c_double),
                                 ctypes.POINTER(ctypes.c_double)]),
    "mpf_point_sample": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    "mpf_mask_loss_forward": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp,
                                       _c_vp, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_pack_mask_bits": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "mpf_mask_loss_backward": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_vp,
                                        _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp]),
    "mpf_select_uncertain": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_sample_select_uncertain": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_match_cost": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                _c_int, _c_int, _c_int, ctypes.c_float, ctypes.c_float, _c_int, _c_vp]),
    "mpf_attn_mask": (_c_int, [_c_vp, _c_int, ctypes.c_int64, ctypes.c_int64, _c_int, _c_int, _c_vp, _c_int, _c_vp,
                               _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_attn_workspace_bytes": (ctypes.c_size_t, [_c_int] * 4),
    "mpf_attn_forward": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                  ctypes.c_float, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_attn_backward": (_c_int, [_c_vp] * 8 + [_c_int] + [_c_vp] * 5 + [_c_int] * 6 + [ctypes.c_float, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_attn_transpose2": (_c_int, [_c_vp] * 4 + [_c_int] * 4 + [_c_vp]),
    "mpf_attn_transpose2_strided": (_c_int, [_c_vp] * 2 + [ctypes.c_int64] * 2 + [_c_vp] * 2 + [_c_int] * 4 + [_c_vp]),
    "mpf_attn_forward_kv": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, ctypes.c_int64, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_int, _c_int,
                                     _c_int, _c_int, _c_int, ctypes.c_float, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_attn_backward_kv": (_c_int, [_c_vp] * 3 + [ctypes.c_int64] * 2 + [_c_vp] * 5 + [_c_int] + [_c_vp] * 5 + [ctypes.c_int64] * 2
                             + [_c_int] * 6 + [ctypes.c_float, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_attn_backward_kv_aux": (_c_int, [_c_vp] * 3 + [ctypes.c_int64] * 2 + [_c_vp] * 5 + [_c_int] + [_c_vp] * 5 + [ctypes.c_int64] * 2
                                 + [_c_int] * 6 + [ctypes.c_float, _c_vp, ctypes.c_size_t, _c_vp, _c_vp]),
    "mpf_attn_bwd_aux_bytes": (ctypes.c_size_t, [_c_int] * 5),
    "mpf_attn_bwd_prep_aux": (_c_int, [_c_vp] * 5 + [_c_int] * 2 + [_c_vp] * 4 + [ctypes.c_size_t] + [_c_int] * 4 + [_c_vp]),
    "mpf_attn_delta": (_c_int, [_c_vp] * 3 + [_c_int] * 3 + [_c_vp]),
    "mpf_attn_bwd_prep": (_c_int, [_c_vp] * 6 + [_c_int] * 4 + [_c_vp]),
    "mpf_gemm3_split": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "mpf_gemm3_tn": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64,
                           _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_nt": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_int, _c_vp, _c_vp, _c_vp,
                           _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_amax_f32": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, _c_vp]),
    "mpf_amax_f32_grouped": (_c_int, [_c_vp, _c_int, ctypes.c_int64, _c_vp]),
    "mpf_h2_range_stats": (_c_int, [_c_vp, _c_int, _c_int, ctypes.c_int64, _c_vp, _c_int, _c_vp, _c_vp]),
    "mpf_gemm3_split_grouped_h2": (_c_int, [_c_vp, _c_int, ctypes.c_int64, _c_vp]),
    "mpf_gemm3_tn_h2": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64,
                              _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_tn_h2_bits": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64,
                                   _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_vp, ctypes.c_int64,
                                   _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_nt_h2": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, _c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_vp, _c_vp,
                              _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_nt_grouped_h2": (_c_int, [_c_vp, _c_int, _c_int, _c_int, ctypes.c_int64, _c_vp]),
    "mpf_gemm3_conv3x3_h2": (_c_int, [_c_vp] * 7 + [_c_int] * 6 + [_c_vp]),
    "mpf_gemm3_conv3x3_wgrad_h2": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_nt_grouped": (_c_int, [_c_vp, _c_int, _c_int, _c_int, ctypes.c_int64, _c_vp]),
    "mpf_small_gemm_bf16": (_c_int, [_c_vp, ctypes.c_int64, ctypes.c_int64, _c_vp, _c_vp, ctypes.c_int64, ctypes.c_int64, _c_vp, _c_vp,
                                     ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_encoder_fields": (_c_int, []),
    "mpf_encoder_forward": (_c_int, [_c_vp, _c_vp]),
    "mpf_encoder_bwd_fields": (_c_int, []),
    "mpf_encoder_backward": (_c_int, [_c_vp, _c_vp]),
    "mpf_decoder_layer_struct_bytes": (ctypes.c_uint64, [_c_int]),
    "mpf_decoder_layer_scratch_bytes": (ctypes.c_uint64, [_c_int] * 6),
    "mpf_decoder_layer_forward": (_c_int, [_c_vp, _c_vp]),
    "mpf_next_attn_mask_scratch_bytes": (ctypes.c_size_t, [_c_int, _c_int]),
    "mpf_next_attn_mask": (_c_int, [_c_vp, _c_vp]),
    "mpf_lin256_res_ln_forward": (_c_int, [_c_vp] * 11 + [_c_int, ctypes.c_float, _c_vp]),
    "mpf_ln256_mlp3_forward": (_c_int, [_c_vp] * 10 + [_c_int, ctypes.c_float, _c_vp]),
    "mpf_decoder_layer_backward": (_c_int, [_c_vp, _c_vp, _c_vp]),
    "mpf_pool_features": (_c_int, [_c_vp, _c_int, _c_vp] + [_c_int] * 6 + [_c_vp]),
    "mpf_pool_features_cl": (_c_int, [_c_vp, ctypes.c_int64, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_mask_head_bits": (_c_int, [_c_vp, ctypes.c_int64, ctypes.c_int64, _c_vp, _c_vp, _c_int, _c_vp, _c_vp] + [_c_int] * 3 + [_c_vp]),
    "mpf_lsa_assign": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, ctypes.c_int64] + [_c_vp] * 7),
    "mpf_lsa_assign_status": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, ctypes.c_int64] + [_c_vp] * 8),
    "mpf_mask_block_empty": (_c_int, [_c_vp, _c_vp] + [_c_int] * 5 + [_c_vp]),
    "mpf_gemm_nt_bf16_workspace_bytes": (ctypes.c_size_t, [_c_int] * 4),
    "mpf_gemm_nt_bf16": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp,
                                  ctypes.c_size_t, _c_vp]),
    "mpf_gemm3_nt_reduce": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_int, _c_vp, _c_vp, _c_vp]),
    "mpf_grouped_scale_cast": (_c_int, [_c_vp, _c_int, ctypes.c_int64, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_split_grouped": (_c_int, [_c_vp, _c_int, ctypes.c_int64, _c_vp]),
    "mpf_decoder_inputs_forward": (_c_int, [_c_vp, ctypes.c_int64, ctypes.c_int64, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int,
                                            _c_vp]),
    "mpf_decoder_inputs_backward": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, ctypes.c_int64, ctypes.c_int64, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_clip_adamw_step": (_c_int, [_c_vp, _c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     _c_vp, _c_vp, _c_vp]),
    "mpf_transpose_f32": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_small_gemm_bf16_blocked": (_c_int, [_c_vp, ctypes.c_int64, ctypes.c_int64, _c_int, ctypes.c_int64, _c_vp, _c_vp, ctypes.c_int64,
                                             ctypes.c_int64, _c_vp, _c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_int, ctypes.c_int64,
                                             _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_small_gemm_bf16_group": (_c_int, [_c_vp, _c_int, _c_vp]),
    "mpf_transpose_group_bf16": (_c_int, [_c_vp, _c_int, _c_int, _c_vp]),
    "mpf_class_loss_forward": (_c_int, [_c_vp, _c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _c_vp, _c_int, _c_vp, _c_int, _c_int,
                                        _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "mpf_class_loss_backward": (_c_int, [_c_vp, _c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _c_vp, _c_int, _c_vp, _c_int, _c_int,
                                         _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "mpf_mask_loss_finalize": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "mpf_mask_loss_finalize_backward": (_c_int, [_c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    "mpf_bias_act": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_relu_bwd_add": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_int, _c_vp]),
    "mpf_maxpool3x3s2_forward": (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    "mpf_maxpool3x3s2_backward": (_c_int, [_c_vp] * 3 + [_c_int] * 4 + [_c_vp]),
    "mpf_upload_small": (_c_int, [_c_vp, _c_vp, ctypes.c_int64, _c_vp]),
    "mpf_tall_gemm_bf16": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_vp, ctypes.c_int64, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_res_ln256_forward": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, ctypes.c_float,
                                       _c_vp, _c_int, _c_vp, _c_vp]),
    "mpf_res_ln256_forward_b": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, ctypes.c_float,
                                         _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "mpf_res_ln256_backward_ws_amax": (_c_int, [_c_vp] * 11 + [_c_int, _c_vp, ctypes.c_size_t, _c_vp, _c_vp]),
    "mpf_res_ln256_backward": (_c_int, [_c_vp] * 11 + [_c_int, _c_vp]),
    "mpf_mask_loss_backward_dense": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_vp,
                                              _c_vp, _c_int, _c_vp, _c_int, _c_int, _c_vp]),
    "mpf_gn_cl_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "mpf_gn_cl_workspace_bytes": (ctypes.c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "mpf_gn_cl_forward": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, ctypes.c_float, _c_int,
                                   _c_vp, ctypes.c_int64, _c_int, _c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_gn_cl_backward": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int,
                                    _c_int, _c_int, _c_vp, ctypes.c_int64, _c_vp, _c_vp, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_upsample2x_cl_backward": (_c_int, [_c_vp, ctypes.c_int64, _c_int, _c_int, _c_int, _c_int, _c_vp, ctypes.c_int64, _c_vp]),
    "mpf_gemm3_conv3x3": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_conv3x3_wgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_nt_reduce_levels": (_c_int, [_c_vp, ctypes.c_int64, _c_vp, ctypes.c_int64, _c_int, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "mpf_gemm3_tn_ex": (_c_int, [_c_vp, _c_int, ctypes.c_int64, _c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_vp, _c_int, ctypes.c_int64,
                                 _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_gemm3_nt_ex": (_c_int, [_c_vp, _c_int, ctypes.c_int64, _c_vp, _c_int, ctypes.c_int64, _c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mpf_res_ln256_backward_workspace_bytes": (ctypes.c_size_t, [_c_int]),
    "mpf_res_ln256_backward_partial": (_c_int, [_c_vp] * 9 + [_c_int, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_res_ln256_backward_partial_amax": (_c_int, [_c_vp] * 9 + [_c_int, _c_vp, ctypes.c_size_t, _c_vp, _c_vp]),
    "mpf_ln_partial_reduce": (_c_int, [_c_vp, ctypes.c_size_t, _c_int, _c_int, _c_vp, _c_vp]),
    "mpf_res_ln256_backward_det_workspace_bytes": (ctypes.c_size_t, [_c_int]),
    "mpf_res_ln256_backward_det": (_c_int, [_c_vp] * 10 + [_c_int, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_res_ln256_backward_ws": (_c_int, [_c_vp] * 11 + [_c_int, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_group_stats_workspace_bytes": (ctypes.c_size_t, [_c_int, ctypes.c_int64]),
    "mpf_group_stats": (_c_int, [_c_vp, _c_int, ctypes.c_int64, ctypes.c_float, _c_vp, _c_vp, _c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_msda_forward": (_c_int, [_c_vp] * 6 + [_c_int] * 8 + [_c_vp]),
    "mpf_msda_backward": (_c_int, [_c_vp] * 9 + [_c_int] * 8 + [_c_vp]),
    "mpf_msda_forward_hs": (_c_int, [_c_vp] * 7 + [_c_int] * 8 + [_c_vp]),
    "mpf_msda_forward_raw_hs": (_c_int, [_c_vp] * 9 + [_c_int] * 8 + [_c_vp]),
    "mpf_msda_forward_raw": (_c_int, [_c_vp] * 8 + [_c_int] * 8 + [_c_vp]),
    "mpf_msda_backward_ws_raw": (_c_int, [_c_vp] * 7 + [_c_int] * 8 + [_c_vp, ctypes.c_size_t, _c_vp]),
    "mpf_msda_backward_ws_raw_o": (_c_int, [_c_vp] * 8 + [_c_int] * 8 + [_c_vp, ctypes.c_size_t, _c_vp, _c_vp, _c_vp]),
    "mpf_msda_backward_workspace_bytes": (ctypes.c_size_t, [_c_int] * 5 + [_c_vp]),
    "mpf_msda_backward_ws": (_c_int, [_c_vp] * 8 + [_c_int] * 8 + [_c_vp, ctypes.c_size_t, _c_vp]),
}


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """The loaded library (raises NativeLibraryError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mp_former_amd/csrc`.  mp_former_amd has no CPU / eager fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        v = l.mpf_abi_version()
        if v != ABI_VERSION:
            raise NativeLibraryError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
        _lib = l
        # MPF_OPTIONS="key=value,key=value": mpf_set_option at load (A/B runs of a kernel switch without code changes)
        for kv in filter(None, os.environ.get("MPF_OPTIONS", "").split(",")):
            k, _, v = kv.partition("=")
            set_option(k.strip(), int(v))
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().mpf_last_error().decode()
        raise RuntimeError(f"{what} failed with code {code}: {msg}")


_option_gen = 0


def option_generation():
    """advanced by every set_option: caches of option-dependent sizes (workspace bytes) key on it"""
    return _option_gen


def set_option(key, value):
    global _option_gen
    check(lib().mpf_set_option(key.encode(), int(value)), f"mpf_set_option({key})")
    _option_gen += 1


def msda_stats(reset=True):
    """Route counters of the blocked MSDA kernels (tests; enable with set_option("msda_stats", 1)) -> dict."""
    out = (ctypes.c_ulonglong * 9)()
    check(lib().mpf_msda_stats(out, 9, 1 if reset else 0), "mpf_msda_stats")
    keys = ("fwd_lds", "fwd_gather", "push_lds", "push_gather", "push_direct", "push_hash", "spill_entries", "pull_split", "pull_single")
    return dict(zip(keys, (int(v) for v in out)))


def last_kernel():
    return lib().mpf_last_kernel().decode()


def profile_enable(on=True):
    check(lib().mpf_profile_enable(1 if on else 0), "mpf_profile_enable")


def profile_get_flops(name_substr):
    """-> summed algorithmic flops of the logged launches matching the name."""
    fl = ctypes.c_double(0)
    check(lib().mpf_profile_get_flops(name_substr.encode(), ctypes.byref(fl)), "mpf_profile_get_flops")
    return fl.value


def profile_get(name_substr):
    """-> (launch count, total ms, total algorithmic bytes) of the logged launches matching the name."""
    n, ms, by = _c_int(0), ctypes.c_double(0), ctypes.c_double(0)
    check(lib().mpf_profile_get(name_substr.encode(), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)),
          "mpf_profile_get")
    return n.value, ms.value, by.value


# ---- workspace scopes -------------------------------------------------------------------------------------------------------
# The python mirrors keep grow-on-demand scratch buffers per device (MSDA entry runs, GroupNorm / LayerNorm partials, decoder
# scratch).  A HIP graph bakes the address of the buffer it was captured with: if eager code later asks the same cache for more
# bytes, the old buffer is freed and every replay writes through a dangling pointer (measured: memory fault in the first
# step after capturing the pixel decoder).  Code that is being captured therefore runs inside ``workspace_scope(name)``, which
# is part of every cache key: buffers baked into a graph are private to it and never resized by anyone else.
_WS_SCOPE = [""]


def ws_scope():
    return _WS_SCOPE[-1]


class workspace_scope:
    def __init__(self, name):
        self.name = str(name)

    def __enter__(self):
        _WS_SCOPE.append(self.name)
        return self

    def __exit__(self, *exc):
        _WS_SCOPE.pop()
        return False


def stream_ptr(device):
    """Raw hipStream_t (as an int) of torch's current stream on ``device``.  ``torch.cuda.current_stream(device).cuda_stream`` builds
    a Stream object through two python-level device-index normalisations (~4.5 us; ~500 calls per training step)."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


class _NullGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_GUARD = _NullGuard()


def device_guard(device):
    """``with torch.cuda.device(device)`` only when ``device`` is not the current one (the common single-device-per-process case
    costs one C call instead of a context manager with two index normalisations)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NULL_GUARD
    return torch.cuda.device(device)
