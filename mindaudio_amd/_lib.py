"""ctypes binding of libmindaudio_amd.so (the C-ABI declared in include/mindaudio_amd.h).

There is NO CPU fallback: if the library is missing, or a call returns an error code, this module
raises.  The only thing resolved without the library is the symbol table used by the CPU tests.
"""
import ctypes
import os
import threading

from . import _build

ABI_VERSION = 3  # MA_ABI_VERSION of include/mindaudio_amd.h this binding was written against (3: round-5, see the header)
MA_OK = 0
MA_ERR_INVALID_ARG = -1
MA_ERR_NFFT_TOO_LARGE = -2
MA_ERR_HOP = -3
MA_ERR_WINDOW = -4
MA_ERR_UNSUPPORTED = -5
MA_ERR_LAUNCH = -6
MA_ERR_WORKSPACE = -7

PAD_MODES = {"constant": 0, "reflect": 1, "edge": 2, "symmetric": 3}
STFT_FRAME_MAJOR = 0
STFT_FREQ_MAJOR = 1

c_f32p = ctypes.c_void_p  # device pointers travel as integers
i64 = ctypes.c_int64
i32 = ctypes.c_int32
f32 = ctypes.c_float
u32 = ctypes.c_uint32
vp = ctypes.c_void_p


class MelBank(ctypes.Structure):
    """struct ma_melbank (include/mindaudio_amd.h)."""

    _fields_ = [
        ("n_mels", i32),
        ("n_freqs", i32),
        ("n_rows", i32),
        ("total_steps", i32),
        ("steps", ctypes.c_void_p),
        ("row_off", ctypes.c_void_p),
        ("start", ctypes.c_void_p),
        ("weights", ctypes.c_void_p),
    ]


class TransposeItem(ctypes.Structure):
    """struct ma_transpose_item (include/mindaudio_amd.h)."""

    _fields_ = [("inp", ctypes.c_void_p), ("out", ctypes.c_void_p), ("ld_in", ctypes.c_int64), ("ld_out", ctypes.c_int64),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("first_block", ctypes.c_int32), ("tiles_c", ctypes.c_int32)]


class ReduceItem(ctypes.Structure):
    """struct ma_reduce_item (include/mindaudio_amd.h)."""

    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("mn", ctypes.c_int64), ("ldo", ctypes.c_int64),
                ("N", ctypes.c_int32), ("splits", ctypes.c_int32), ("alpha", ctypes.c_float), ("accumulate", ctypes.c_int32),
                ("first_block", ctypes.c_int32), ("pstride", ctypes.c_int32)]


class TrainEpilogue(ctypes.Structure):
    """struct ma_train_epilogue (include/mindaudio_amd.h)."""

    _fields_ = [("mode", ctypes.c_int32), ("ln_out_bf16", ctypes.c_int32), ("bias", ctypes.c_void_p), ("aux", ctypes.c_void_p),
                ("ld_aux", ctypes.c_int64), ("out2", ctypes.c_void_p), ("ldo2", ctypes.c_int64), ("residual", ctypes.c_void_p),
                ("ldr", ctypes.c_int64), ("row_scale", ctypes.c_void_p), ("alpha", ctypes.c_float), ("p", ctypes.c_float),
                ("seed", ctypes.c_uint32), ("salt", ctypes.c_uint32), ("ln_gamma1", ctypes.c_void_p), ("ln_beta1", ctypes.c_void_p),
                ("ln_gamma2", ctypes.c_void_p), ("ln_beta2", ctypes.c_void_p), ("ln_row_scale", ctypes.c_void_p),
                ("ln_out", ctypes.c_void_p), ("ln_mid", ctypes.c_void_p), ("ld_ln", ctypes.c_int64), ("ld_mid", ctypes.c_int64),
                ("ln_eps", ctypes.c_float), ("act", ctypes.c_int32)]


class TnItem(ctypes.Structure):
    """struct ma_tn_item (host array: the arguments of one ma_gemm_tn_partial_bf16 call each)."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("partial", ctypes.c_void_p), ("lda", ctypes.c_int64),
                ("ldb", ctypes.c_int64), ("Mo", ctypes.c_int64), ("No", ctypes.c_int64), ("Kc", ctypes.c_int64),
                ("Mo_store", ctypes.c_int64), ("partial_bytes", ctypes.c_int64), ("with_colsum", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class TnDirectItem(ctypes.Structure):
    """struct ma_tn_direct_item (host array: one weight-gradient product each, no split-K)."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("out", ctypes.c_void_p), ("colsum", ctypes.c_void_p),
                ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldo", ctypes.c_int64), ("Mo", ctypes.c_int32),
                ("No", ctypes.c_int32), ("Kc", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class PackItem(ctypes.Structure):
    """struct ma_pack_item (include/mindaudio_amd.h)."""

    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("ld", ctypes.c_int64), ("N", ctypes.c_int32),
                ("K", ctypes.c_int32), ("kind", ctypes.c_int32), ("first_block", ctypes.c_int32)]


class GemmEpilogue(ctypes.Structure):
    """struct ma_gemm_epilogue (include/mindaudio_amd.h)."""

    _fields_ = [
        ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p),
        ("row_scale", ctypes.c_void_p),
        ("ldr", i64),
        ("alpha", f32),
        ("act", i32),
        ("out_bf16", i32),
        ("col_scale", ctypes.c_void_p),
        ("col_shift", ctypes.c_void_p),
        ("act2", i32),
        ("reserved", i32),
    ]


ACT_NONE, ACT_SWISH, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4

# name -> (restype, argtypes); must list every symbol include/mindaudio_amd.h declares
PROTOTYPES = {
    "ma_abi_version": (ctypes.c_int, []),
    "ma_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "ma_init": (ctypes.c_int, []),
    "ma_init_kernel_attributes": (i32, []),
    "ma_valu_issue_probe": (ctypes.c_int, [i32, i32, c_f32p, vp]),
    "ma_weight_stream_probe": (ctypes.c_int, [vp, i64, i32, c_f32p, vp]),
    "ma_num_frames": (i64, [i64, i32, i32, i32]),
    "ma_mel_row_stride": (i32, [i32]),
    "ma_stft_f32": (ctypes.c_int, [c_f32p, i64, i64, i64, i32, i32, c_f32p, i32, i32, i32, c_f32p, ctypes.c_void_p]),
    "ma_fbank_workspace_bytes": (i64, [i64, i64]),
    "ma_fbank_db_f32": (ctypes.c_int, [c_f32p, i64, i64, i64, i32, i32, c_f32p, i32, i32, ctypes.POINTER(MelBank),
                                       f32, f32, f32, f32, f32, c_f32p, ctypes.c_void_p, i64, ctypes.c_void_p]),
    "ma_melspectrogram_f32": (ctypes.c_int, [c_f32p, i64, i64, i64, i32, i32, c_f32p, i32, i32,
                                             ctypes.POINTER(MelBank), f32, c_f32p, ctypes.c_void_p]),
    "ma_fbank_kaldi_f32": (ctypes.c_int, [c_f32p, ctypes.c_void_p, i64, i64, i64, i32, i32, i32, c_f32p,
                                          ctypes.POINTER(MelBank), f32, c_f32p, ctypes.c_void_p, ctypes.c_void_p, i64,
                                          ctypes.c_void_p]),
    "ma_gemm_bf16": (ctypes.c_int, [ctypes.c_void_p, i64, ctypes.c_void_p, i64, ctypes.c_void_p, i64, i64, i64, i64,
                                    ctypes.POINTER(GemmEpilogue), ctypes.c_void_p]),
    "ma_conv2d_3x3s2_nhwc_bf16": (ctypes.c_int, [ctypes.c_void_p, i64, i64, i64, i64, ctypes.c_void_p, i64,
                                                 ctypes.c_void_p, ctypes.POINTER(GemmEpilogue), ctypes.c_void_p]),
    "ma_layernorm_f32": (ctypes.c_int, [ctypes.c_void_p, i64, i64, i64, ctypes.c_void_p, ctypes.c_void_p, f32,
                                        ctypes.c_void_p, ctypes.c_void_p, i64, i32, ctypes.c_void_p]),
    "ma_layernorm2_f32": (ctypes.c_int, [ctypes.c_void_p, i64, i64, i64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p, f32, ctypes.c_void_p, i64, ctypes.c_void_p,
                                         i64, i32, ctypes.c_void_p]),
    "ma_subsample_conv1_strided_nhwc": (ctypes.c_int, [vp, i64, i64, i64, i64, i64, i32, vp, vp, vp, vp, i32, vp, vp]),
    "ma_subsample_conv1_nhwc": (ctypes.c_int, [ctypes.c_void_p, i64, i64, i32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, i32, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "ma_relpos_attention_workspace_bytes": (i64, [i64, i64, i32, i32]),
    "ma_relpos_attention_bf16": (ctypes.c_int, [ctypes.c_void_p, i64, ctypes.c_void_p, i64, ctypes.c_void_p,
                                                ctypes.c_void_p, ctypes.c_void_p, i64, i64, i32, i32,
                                                ctypes.c_void_p, i64, ctypes.c_void_p, i64, ctypes.c_void_p]),
    "ma_relpos_attention_qmask_bf16": (ctypes.c_int, [ctypes.c_void_p, i64, ctypes.c_void_p, i64, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p, i64, i64, i32, i32,
                                                      ctypes.c_void_p, i64, ctypes.c_void_p, i64, ctypes.c_void_p]),
    "ma_convmodule_mid_bf16": (ctypes.c_int, [ctypes.c_void_p, i64, i64, i64, i32, ctypes.c_void_p, i32,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, i64,
                                              ctypes.c_void_p]),
    "ma_ctc_loss_f32": (ctypes.c_int, [ctypes.c_void_p, i64, i64, i64, i32, ctypes.c_void_p, i32, ctypes.c_void_p,
                                       ctypes.c_void_p, i32, i32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    "ma_ctc_grad_workspace_bytes": (i64, [i64, i64, i32]),
    "ma_ctc_loss_grad_f32": (ctypes.c_int, [vp, i64, i64, i64, i32, vp, i32, vp, vp, i32, i32, f32, vp, vp, vp, vp, i64,
                                            vp, i64, vp]),
    "ma_ctc_greedy_search_f32": (ctypes.c_int, [vp, i64, i64, i64, i32, vp, i32, vp, vp, vp, vp, vp]),
    "ma_cast_f32_bf16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, i64, ctypes.c_void_p]),
    "ma_ecapa_pack_input_bf16": (ctypes.c_int, [vp, i64, i64, i32, i32, i32, vp, vp]),
    "ma_add_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, vp]),
    "ma_time_mean_bf16": (ctypes.c_int, [vp, i64, i64, i64, i32, i32, vp, vp]),
    "ma_se_apply_bf16": (ctypes.c_int, [vp, i64, vp, vp, i64, vp, i64, i64, i64, i32, i32, vp]),
    "ma_se_block_bf16": (ctypes.c_int, [vp, i64, vp, vp, vp, vp, vp, i64, vp, i64, i64, i64, i32, i32, i32, vp]),
    "ma_se_gate_bf16": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "ma_linear_small_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, vp]),
    "ma_asp_fused_bf16": (ctypes.c_int, [vp, i64, vp, vp, i64, i64, i64, i32, i32, i32, f32, vp, vp, vp, vp]),
    "ma_asp_pool_bf16": (ctypes.c_int, [vp, i64, vp, i64, i64, i64, i32, i32, f32, vp, vp, vp, vp]),
    "ma_compute_deltas_f32": (ctypes.c_int, [vp, i64, i64, i32, i32, vp, vp]),
    "ma_context_window_f32": (ctypes.c_int, [vp, i64, i32, i64, i32, i32, vp, vp]),
    "ma_dct_f32": (ctypes.c_int, [vp, i64, i32, i64, vp, i32, vp, vp]),
    "ma_magphase_f32": (ctypes.c_int, [vp, i64, f32, vp, vp, vp]),
    "ma_magphase_angle_f32": (ctypes.c_int, [vp, i64, f32, vp, vp, vp]),
    "ma_pointwise_f32": (ctypes.c_int, [vp, i64, i32, f32, f32, vp, vp]),
    "ma_frame_f64": (ctypes.c_int, [vp, i32, i64, i64, i64, i32, i32, vp, vp]),
    "ma_cmvn_stats_f64": (ctypes.c_int, [vp, vp, i64, i64, i32, vp, vp]),
    "ma_subsampled_mask_len": (i32, [i32]),
    "ma_collate_asr_i32": (ctypes.c_int, [ctypes.c_void_p] * 3 + [i32] * 7 + [ctypes.c_void_p] * 11),
    "ma_wave_rows_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, i64, vp]),
    "ma_spec_aug_f32": (ctypes.c_int, [ctypes.c_void_p, i64, i64, i32, ctypes.c_void_p, ctypes.c_void_p, i32,
                                       ctypes.c_void_p, i32, ctypes.c_void_p]),
    "ma_conv1d_taps_bf16": (ctypes.c_int, [vp, i64, i64, i64, i32, i32, vp, vp, i64, i64, ctypes.POINTER(GemmEpilogue),
                                           vp]),
    "ma_gemm_splitk_workspace_bytes": (i64, [i64, i64, i64]),
    "ma_gemm_bf16_splitk_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, i64, f32, i32, vp, i64, vp]),
    "ma_gemm_bf16_splitk_join_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, i64, ctypes.POINTER(TrainEpilogue), vp, i64, vp]),
    "ma_gemm_tn_workspace_bytes": (i64, [i64, i64, i64]),
    "ma_gemm_tn_bf16_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, i64, i64, i64, i64, f32, i32, vp, vp, i64, vp]),
    "ma_conv2d_3x3s2_dw_bf16": (ctypes.c_int, [vp, i64, vp, i64, i64, i64, i64, i64, vp, vp, vp, i64, vp]),
    "ma_conv2d_3x3s2_dw_workspace_bytes": (i64, [i64, i64, i64]),
    "ma_transpose_bf16": (ctypes.c_int, [vp, i64, i64, i64, vp, i64, vp, vp]),
    "ma_transpose_batch_bf16": (ctypes.c_int, [vp, vp, i32, vp]),
    "ma_gemm_tn_splits": (i32, [i64, i64, i64]),
    "ma_gemm_tn_partial_bf16": (ctypes.c_int, [vp, i64, vp, i64, i64, i64, i64, i64, i32, vp, i64, vp]),
    "ma_reduce_splits_batch_f32": (ctypes.c_int, [vp, vp, i32, vp]),
    "ma_layernorm_bwd_f32": (ctypes.c_int, [vp, i64, i64, i64, vp, f32, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, vp]),
    "ma_train_reduce_workspace_bytes": (i64, []),
    "ma_layernorm_bwd_next_f32": (ctypes.c_int, [vp, i64, i64, i64, vp, f32, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, vp, i64,
                                                 f32, vp, f32, u32, u32, vp]),
    "ma_gemm_k256_train_bf16": (ctypes.c_int, [vp, i64, vp, vp, i64, i64, i64, ctypes.POINTER(TrainEpilogue), vp]),
    "ma_gemm_rows_train_bf16": (ctypes.c_int, [vp, i64, i64, i64, vp, vp, i64, ctypes.POINTER(TrainEpilogue), vp]),
    "ma_gemm_rows_train_parts": (i32, [i64]),
    "ma_ffn_train_rows": (i32, []),
    "ma_ffn_train_parts": (i32, [i64]),
    "ma_ffn_train_bf16": (ctypes.c_int, [vp, i64, i64, i32, vp, vp, vp, vp, i64, i32, f32, u32, u32, vp, i64, ctypes.POINTER(TrainEpilogue), vp]),
    "ma_ffn_train_bwd_bf16": (ctypes.c_int, [vp, i64, i64, i32, vp, vp, vp, i64, vp, i64, ctypes.POINTER(TrainEpilogue),
                                             ctypes.POINTER(TrainEpilogue), vp]),
    "ma_conv2d_3x3s2_dinput_bf16": (ctypes.c_int, [vp, i64, i64, i64, i64, vp, vp, vp, vp, vp]),
    "ma_gemm_tn_partial_group_bf16": (ctypes.c_int, [ctypes.POINTER(TnItem), i32, vp]),
    "ma_debug_tn_group_lds": (ctypes.c_int, [i32]),
    "ma_gemm_tn_direct_max_items": (i32, []),
    "ma_gemm_tn_direct_group_bf16": (ctypes.c_int, [ctypes.POINTER(TnDirectItem), i32, vp]),
    "ma_convmid_bwd_parts": (i32, [i64, i64]),
    "ma_pack_item_pieces": (i64, [i32, i64, i64]),
    "ma_pack_batch_bf16": (ctypes.c_int, [vp, vp, i32, vp]),
    "ma_act_dropout_fwd_bf16": (ctypes.c_int, [vp, vp, i64, i32, f32, u32, u32, vp]),
    "ma_act_dropout_bwd_bf16": (ctypes.c_int, [vp, vp, vp, i64, i32, f32, u32, u32, vp]),
    "ma_dropout_add_f32": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, i32, i64, i64, f32, f32, u32, u32, vp]),
    "ma_dropout_bwd_bf16": (ctypes.c_int, [vp, i64, vp, i64, i64, i64, f32, vp, f32, u32, u32, vp]),
    "ma_convmid_fwd_train": (ctypes.c_int, [vp, i64, i64, i64, i32, vp, i32, vp, vp, vp, vp]),
    "ma_bn_finalize_f32": (ctypes.c_int, [vp, i32, i32, i64, f32, f32, vp, vp, vp, vp]),
    "ma_convmid_fwd_train_parts": (i32, [i64, i64, i32]),
    "ma_layernorm_bwd_parts": (i32, [i64]),
    "ma_bn_swish_fwd_bf16": (ctypes.c_int, [vp, vp, vp, vp, vp, i64, i32, vp]),
    "ma_bn_swish_bwd_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, i64, vp]),
    "ma_convmid_bwd_bf16": (ctypes.c_int, [vp, vp, i64, i64, i64, i32, vp, i32, vp, i64, vp, vp, vp, i64, vp]),
    "ma_bn_swish_bwd_stage1_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, i64, vp]),
    "ma_convmid_bwd_bn_bf16": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, vp, i32, vp, i64, vp, vp, vp, i64, vp]),
    "ma_block_table_create": (vp, []),
    "ma_block_table_destroy": (None, [vp]),
    "ma_block_table_entry_point": (i32, [ctypes.c_char_p]),
    "ma_block_table_entry_point_params": (i32, [i32]),
    "ma_block_table_entry_point_seeds": (ctypes.c_uint64, [i32]),
    "ma_block_table_add": (ctypes.c_int, [vp, i32, i32, i32, ctypes.POINTER(i64), i32, vp, i64]),
    "ma_block_table_calls": (i32, [vp, i32, i32]),
    "ma_block_table_failed_call": (i32, [vp]),
    "ma_block_table_call_entry_point": (i32, [vp, i32, i32, i32]),
    "ma_block_table_call_word": (i64, [vp, i32, i32, i32, i32]),
    "ma_block_table_call_blob": (i64, [vp, i32, i32, i32, vp, i64]),
    "ma_conformer_block_fwd_train": (ctypes.c_int, [vp, i32, u32, vp]),
    "ma_conformer_block_bwd_train": (ctypes.c_int, [vp, i32, u32, vp]),
    "ma_relu_bwd_bf16": (ctypes.c_int, [vp, vp, i64, vp]),
    "ma_im2col_t_3x3s2_nhwc_bf16": (ctypes.c_int, [vp, i64, i64, i64, i64, vp, i64, vp]),
    "ma_col2im_3x3s2_relu_bf16": (ctypes.c_int, [vp, vp, i64, i64, i64, i64, vp, vp]),
    "ma_subsample_conv1_dw_f32": (ctypes.c_int, [vp, vp, i64, i64, i32, vp, vp, i32, vp, vp, vp, i64, vp]),
    "ma_relpos_attention_train_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i32, i32, vp, i64, vp, i64,
                                                      vp, vp]),
    "ma_relpos_attention_bwd_workspace_bytes": (i64, [i64, i64, i32, i32]),
    "ma_relpos_attention_bwd_layout": (ctypes.c_int, [i64, i64, i32, i32, ctypes.POINTER(i64), ctypes.POINTER(i64),
                                                      ctypes.POINTER(i32), ctypes.POINTER(i32)]),
    "ma_relpos_attention_bwd_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i32,
                                                    i32, vp, i64, vp, i64, vp, vp, vp, i64, vp]),
    "ma_relpos_attention_train_qmask_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i32, i32, vp, i64, vp,
                                                            i64, vp, vp]),
    "ma_relpos_attention_bwd_qmask_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i32,
                                                          i32, vp, i64, vp, i64, vp, vp, vp, i64, vp]),
    "ma_embed_posenc_f32": (ctypes.c_int, [vp, vp, vp, i64, i32, i32, i32, f32, f32, u32, u32, vp, vp]),
    "ma_embed_bwd_f32": (ctypes.c_int, [vp, vp, i64, i32, i32, f32, f32, u32, u32, vp, vp]),
    "ma_embed_bwd_rows_f32": (ctypes.c_int, [vp, vp, vp, i64, i32, i32, f32, f32, u32, u32, vp, vp]),
    "ma_mha_small_fwd_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, i32, i64, i32, i32, i32, i32, f32, vp, i64, vp,
                                             vp]),
    "ma_mha_small_bwd_bf16": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i32, i32, i32, i32, f32,
                                             vp, i64, vp, i64, vp, i64, vp]),
    "ma_label_smoothing_loss_grad_len_f32": (ctypes.c_int, [vp, i64, i64, i32, vp, vp, f32, f32, vp, vp, i64, vp, vp, vp]),
    "ma_label_smoothing_loss_grad_len_x32": (ctypes.c_int, [vp, i64, i64, i32, vp, vp, f32, f32, vp, vp, i64, vp, vp, vp]),
    "ma_mha_small_fwd_x32": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, i32, i64, i32, i32, i32, i32, f32, vp, i64, vp, vp]),
    "ma_mha_small_bwd_x32": (ctypes.c_int, [vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i32, i32, i32, i32, f32,
                                            vp, i64, vp, i64, vp, i64, vp]),
    "ma_resample_fft_length": (i64, [i64, i64]),
    "ma_resample_fft_workspace_bytes": (i64, [i64, i64, i64]),
    "ma_resample_fft_f32": (ctypes.c_int, [vp, i64, vp, vp, i64, i64, i64, vp, i64, vp, i64, vp]),
    "ma_fft_pow2_c32": (ctypes.c_int, [vp, vp, i64, i64, i32, ctypes.POINTER(ctypes.c_void_p), vp]),
    "ma_istft_workspace_bytes": (i64, [i64, i64, i32]),
    "ma_istft_f32": (ctypes.c_int, [vp, i64, i32, i64, i64, i32, vp, i32, vp, i64, vp, i64, vp]),
    "ma_conv2d_3x3s2_packed_bytes": (i64, [i64, i64]),
    "ma_conv2d_3x3s2_pack_bf16": (ctypes.c_int, [vp, i64, i64, vp, vp]),
    "ma_conv2d_3x3s2_packed_nhwc_bf16": (ctypes.c_int, [vp, i64, i64, i64, i64, vp, i64, vp, i32, vp, vp]),
    "ma_subsample_fused_packed_bytes": (i64, [i64, i64]),
    "ma_subsample_fused_pack_bf16": (ctypes.c_int, [vp, vp, i64, i64, vp, vp]),
    "ma_subsample_fused_bf16": (ctypes.c_int, [vp, i64, i64, i64, i64, i64, i32, vp, vp, vp, vp, vp, i64, vp, vp]),
    "ma_gemm_rows_packed_bytes": (i64, [i64, i64]),
    "ma_gemm_rows_pack_bf16": (ctypes.c_int, [vp, i64, i64, i64, vp, vp]),
    "ma_gemm_rows_packed_f32": (ctypes.c_int, [vp, i64, i64, i64, vp, i64, vp, f32, vp, i64, vp]),
    "ma_convmid_pw2_bf16": (ctypes.c_int, [vp, i64, i64, i64, i32, vp, i32, vp, vp, vp, vp, vp, vp, i64, vp]),
    "ma_convmodule_bf16": (ctypes.c_int, [vp, i64, i64, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i64, vp]),
    "ma_attn_out_convmodule_bf16": (ctypes.c_int, [vp, i64, vp, vp, vp, vp, f32, i64, i64, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp,
                                                   vp, vp, i64, vp]),
    "ma_gemm_k256_packed_bytes": (i64, [i64, i64]),
    "ma_gemm_k256_pack_bf16": (ctypes.c_int, [vp, i64, i64, i64, vp, vp]),
    "ma_gemm_k256_packed_bf16": (ctypes.c_int, [vp, i64, vp, vp, i64, i64, i64, i64, vp, vp]),
    "ma_gemm_k256_packed_ln_bf16": (ctypes.c_int, [vp, i64, vp, vp, i64, i64, i64, i64, vp, vp, vp, f32, vp, vp, i64, vp]),
    "ma_ffn_packed_bytes": (i64, [i32, i32]),
    "ma_ffn_pack_weights_bf16": (ctypes.c_int, [vp, vp, i32, i32, vp, vp]),
    "ma_ffn_packed_bf16": (ctypes.c_int, [vp, i64, vp, vp, vp, vp, i64, i64, i32, i32, f32, i32, vp, vp, vp, vp, f32, vp, i64,
                                          i32, vp, vp, vp]),
    "ma_ffn_packed_pair_bf16": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp,
                                               f32, vp, i64, vp]),
    "ma_ffn_qkv_packed_bytes": (i64, [i64]),
    "ma_ffn_qkv_pack_bf16": (ctypes.c_int, [vp, i64, i64, vp, vp]),
    "ma_ffn_packed_qkv_bf16": (ctypes.c_int, [vp, i64, vp, vp, vp, vp, i64, i64, i32, i32, f32, vp, vp, vp, vp, f32, vp, vp, i64, vp,
                                              i64, vp]),
    "ma_ffn_packed_pair_qkv_bf16": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp,
                                                   vp, f32, vp, vp, i64, vp, i64, vp]),
    "ma_grad_overflow_f32": (ctypes.c_int, [vp, i64, vp, vp]),
    "ma_adam_f32": (ctypes.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp]),
    "ma_adam_mirror_f32": (ctypes.c_int, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, vp, vp]),
    "ma_db_workspace_bytes": (i64, [i64, i64]),
    "ma_amplitude_to_db_f32": (ctypes.c_int, [c_f32p, i64, i64, f32, f32, f32, f32, c_f32p, ctypes.c_void_p, i64,
                                              ctypes.c_void_p]),
    "ma_res2net_fused_lds_bytes": (i64, [i32, i64, i32]),
    "ma_res2net_fused_bf16": (ctypes.c_int, [vp, i64, vp, i64, i64, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    # ---- float32 validation mode (x32) ----
    "ma_gemm_x32": (ctypes.c_int, [vp, i64, i64, vp, i64, i64, vp, i64, i64, i64, i64, ctypes.POINTER(GemmEpilogue), vp]),
    "ma_colsum_x32": (ctypes.c_int, [vp, i64, i64, i64, vp, i32, vp]),
    "ma_relpos_attention_fwd_x32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i32, i32, vp, i64, vp, vp]),
    "ma_relpos_attention_bwd_x32_workspace_bytes": (i64, [i64, i64, i32]),
    "ma_relpos_attention_bwd_x32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i32, i32,
                                                   vp, i64, vp, i64, vp, vp, vp, i64, vp]),
    "ma_relpos_attention_fwd_qmask_x32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, i64, i64, i32, i32, vp, i64, vp, vp]),
    "ma_relpos_attention_bwd_qmask_x32": (ctypes.c_int, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i32, i32,
                                                         vp, i64, vp, i64, vp, vp, vp, i64, vp]),
    "ma_subsample_conv1_nhwc_x32": (ctypes.c_int, [vp, i64, i64, i64, i64, i64, i32, vp, vp, vp, vp, i32, vp, vp]),
    "ma_im2col_3x3s2_nhwc_x32": (ctypes.c_int, [vp, i64, i64, i64, i64, vp, vp]),
    "ma_col2im_3x3s2_relu_x32": (ctypes.c_int, [vp, vp, i64, i64, i64, i64, vp, vp]),
    "ma_relu_bwd_x32": (ctypes.c_int, [vp, vp, i64, vp]),
    "ma_subsample_conv1_dw_x32": (ctypes.c_int, [vp, vp, i64, i64, i32, vp, vp, i32, vp, vp, vp, i64, vp]),
    "ma_act_dropout_fwd_x32": (ctypes.c_int, [vp, vp, i64, i32, f32, u32, u32, vp]),
    "ma_act_dropout_bwd_x32": (ctypes.c_int, [vp, vp, vp, i64, i32, f32, u32, u32, vp]),
    "ma_dropout_bwd_x32": (ctypes.c_int, [vp, i64, vp, i64, i64, i64, f32, vp, f32, u32, u32, vp]),
    "ma_convmid_fwd_train_x32": (ctypes.c_int, [vp, i64, i64, i64, i32, vp, i32, vp, vp, vp, vp]),
    "ma_bn_swish_fwd_x32": (ctypes.c_int, [vp, vp, vp, vp, vp, i64, i32, vp]),
    "ma_bn_swish_bwd_x32": (ctypes.c_int, [vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, i64, vp]),
    "ma_convmid_bwd_x32": (ctypes.c_int, [vp, vp, i64, i64, i64, i32, vp, i32, vp, i64, vp, vp, vp, i64, vp]),
    "ma_ctc_loss_grad_x32": (ctypes.c_int, [vp, i64, i64, i64, i32, vp, i32, vp, vp, i32, i32, f32, vp, vp, vp, vp, i64, vp,
                                            i64, vp]),
}

_lib = None
_tls = threading.local()  # .rec: the block table THIS thread is filling (train/block_table.py); load() then hands out its proxy


def recording():
    """The block table being filled by the calling thread, or None (other threads keep calling the library itself)."""
    return getattr(_tls, "rec", None)


def set_recording(table):
    _tls.rec = table


class MindaudioAmdError(RuntimeError):
    pass


def lib_path():
    # MINDAUDIO_AMD_LIB: developer override used by tools/ to load an instrumented build of the same sources
    return os.environ.get("MINDAUDIO_AMD_LIB") or _build.LIB_PATH


def load():
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        rec = getattr(_tls, "rec", None)
        return _lib if rec is None else rec._proxy
    path = lib_path()
    if not os.path.exists(path):
        raise MindaudioAmdError(
            "libmindaudio_amd.so is missing (%s). Build it with `python -m mindaudio_amd._build` "
            "(or __graft_entry__.build()); there is no CPU fallback." % path)
    # PyTorch-ROCm bundles its own libamdhip64.so.7.  The process must hold ONE HIP runtime: import torch
    # first so that our library binds to the runtime that owns torch's context, streams and allocations
    # (loading ours first pulls /opt/rocm's copy and every launch on a torch stream then fails).
    import torch  # noqa: F401

    lib = ctypes.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ma_abi_version() != ABI_VERSION:
        raise MindaudioAmdError("ABI version mismatch: library %d, binding %d" % (lib.ma_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(status, what):
    """Map a C-ABI status to the exception the reference raises for the same condition."""
    if status == MA_OK:
        return
    msg = load().ma_status_string(int(status)).decode()
    if status in (MA_ERR_NFFT_TOO_LARGE, MA_ERR_HOP, MA_ERR_WINDOW, MA_ERR_INVALID_ARG):
        raise ValueError("%s: %s" % (what, msg))  # spectrum.py:182-187, 295-296, 331-334
    if status == MA_ERR_UNSUPPORTED:
        raise NotImplementedError("%s: %s" % (what, msg))
    raise MindaudioAmdError("%s: %s (status %d)" % (what, msg, status))
