"""ctypes binding of the C-ABI in ``include/mit_hip.h``.

The library is the product: if it cannot be loaded the import of any op raises — there is
no CPU or PyTorch fallback behind these entry points.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

MIT_MAX_TAPS = 64
MIT_ABI_VERSION = 11

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SILU, ACT_SIGMOID, ACT_GELU = range(6)
ACT_POST_FIRST = 0x100
PAD_ZERO, PAD_REFLECT = 0, 1


class MitTensorMap(C.Structure):
    _fields_ = [
        ("base", C.c_void_p),
        ("zs1", C.c_int64),
        ("zs0", C.c_int64),
        ("bs", C.c_int64),
        ("ys", C.c_int64),
        ("xs", C.c_int64),
        ("nhi", C.c_int64),
        ("nsplit", C.c_int32),
        ("_pad", C.c_int32),
    ]


class MitConvGemm(C.Structure):
    _fields_ = [
        ("a", C.c_void_p),
        ("a_zs1", C.c_int64),
        ("a_zs0", C.c_int64),
        ("a_bs", C.c_int64),
        ("a_ys", C.c_int64),
        ("a_xs", C.c_int64),
        ("NB", C.c_int32),
        ("Hi", C.c_int32),
        ("Wi", C.c_int32),
        ("Cin", C.c_int32),
        ("Ho", C.c_int32),
        ("Wo", C.c_int32),
        ("sy", C.c_int32),
        ("sx", C.c_int32),
        ("ntaps", C.c_int32),
        ("pad_mode", C.c_int32),
        ("tap_dy", C.c_int8 * MIT_MAX_TAPS),
        ("tap_dx", C.c_int8 * MIT_MAX_TAPS),
        ("tap_off", C.c_int32 * MIT_MAX_TAPS),
        ("w", C.c_void_p),
        ("w_zs1", C.c_int64),
        ("w_zs0", C.c_int64),
        ("ldw", C.c_int64),
        ("Kw", C.c_int32),
        ("Nw", C.c_int32),
        ("N", C.c_int32),
        ("Z", C.c_int32),
        ("zdiv", C.c_int32),
        ("c", MitTensorMap),
        ("pre", MitTensorMap),
        ("post", MitTensorMap),
        ("scale", C.c_void_p),
        ("bias", C.c_void_p),
        ("act", C.c_int32),
        ("act_alpha", C.c_float),
        ("w_split", C.c_void_p),
        ("ws_zs0", C.c_int64),
        ("dyn", C.c_void_p),
        ("a_dyn", C.c_int64),
        ("c_dyn", C.c_int64),
        ("lut_rows", C.c_void_p),
        ("lut1", C.c_void_p),
        ("lut2", C.c_void_p),
        ("lut_ld", C.c_int64),
    ]


class MitPGemm(C.Structure):
    """Descriptor of ``mit_pgemm`` (include/mit_hip.h): a plain GEMM on operands that arrive as three bf16 planes."""
    _fields_ = [
        ("a_planes", C.c_void_p), ("a_zs", C.c_int64), ("lda", C.c_int64),
        ("w_planes", C.c_void_p), ("w_zs", C.c_int64), ("ldw", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("Z", C.c_int32),
        ("c", C.c_void_p), ("ldc", C.c_int64), ("c_zs", C.c_int64),
        ("pre", C.c_void_p), ("ld_pre", C.c_int64), ("pre_zs", C.c_int64),
        ("post", C.c_void_p), ("ld_post", C.c_int64), ("post_zs", C.c_int64),
        ("c_planes", C.c_void_p), ("ld_cp", C.c_int64), ("cp_zs", C.c_int64),
        ("scale", C.c_void_p), ("bias", C.c_void_p),
        ("act", C.c_int32), ("act_alpha", C.c_float), ("nprod", C.c_int32), ("tile", C.c_int32),
    ]


class MitXposTables(C.Structure):
    _fields_ = [("cos_t", C.c_void_p), ("sin_t", C.c_void_p), ("scale_t", C.c_void_p), ("iscale_t", C.c_void_p),
                ("imax", C.c_int32), ("pmax", C.c_int32)]


class MitLinear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p), ("ldw", C.c_int64), ("K", C.c_int32),
                ("N", C.c_int32), ("Kp", C.c_int32), ("Np", C.c_int32), ("w_split", C.c_void_p)]


class MitOcrDecoderLayer(C.Structure):
    _fields_ = [("ln1_w", C.c_void_p), ("ln1_b", C.c_void_p), ("ln2_w", C.c_void_p), ("ln2_b", C.c_void_p),
                ("ln3_w", C.c_void_p), ("ln3_b", C.c_void_p), ("qkv", MitLinear), ("out", MitLinear), ("q2", MitLinear),
                ("out2", MitLinear), ("ff1", MitLinear), ("ff2", MitLinear)]


class MitOcr48Decoder(C.Structure):
    _fields_ = [("layers", MitOcrDecoderLayer * 5), ("embd", C.c_void_p), ("pred1", MitLinear), ("pred", MitLinear),
                ("color1", MitLinear), ("color_heads", MitLinear), ("xpos", MitXposTables), ("dict_size", C.c_int32),
                ("_pad", C.c_int32)]


class MitOcr48DecodeArgs(C.Structure):
    _fields_ = [("N", C.c_int32), ("L", C.c_int32), ("mem_k", C.c_void_p), ("mem_v", C.c_void_p), ("mem_len", C.c_void_p),
                ("max_seq_length", C.c_int32), ("start_tok", C.c_int32), ("end_tok", C.c_int32), ("max_finished", C.c_int32),
                ("suppress_eos", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("res_tok", C.c_void_p), ("res_len", C.c_void_p), ("res_prob", C.c_void_p), ("res_row", C.c_void_p),
                ("colors", C.c_void_p), ("trace_logits", C.c_void_p), ("trace_hist", C.c_void_p), ("steps_run", C.c_int32),
                ("graph_mode", C.c_int32)]


class MitProfStat(C.Structure):
    _fields_ = [("launches", C.c_int64), ("ms", C.c_double), ("exec_flops", C.c_double), ("alg_flops", C.c_double)]


class MitProfKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("alg_bytes", C.c_double),
                ("alg_flops", C.c_double)]


class MitWarpLine(C.Structure):
    _fields_ = [("minv", C.c_double * 9), ("page", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32), ("cw", C.c_int32),
                ("ch", C.c_int32), ("dw", C.c_int32), ("dh", C.c_int32), ("vertical", C.c_int32), ("out_row", C.c_int32),
                ("_pad", C.c_int32)]


class MitCrfCrop(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class MitDilateJob(C.Structure):
    _fields_ = [("sx", C.c_int32), ("sy", C.c_int32), ("sw", C.c_int32), ("sh", C.c_int32), ("dx", C.c_int32), ("dy", C.c_int32),
                ("dw", C.c_int32), ("dh", C.c_int32), ("k", C.c_int32), ("spitch", C.c_int32), ("src_off", C.c_int64)]


class MitMaskRun(C.Structure):
    _fields_ = [("y", C.c_int32), ("x0", C.c_int32), ("x1", C.c_int32), ("comp", C.c_int32)]


class MitRefineWindow(C.Structure):
    _fields_ = [("x1", C.c_int32), ("y1", C.c_int32), ("x2", C.c_int32), ("y2", C.c_int32)]


class MitRefineCand(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lo", C.c_int32), ("hi", C.c_int32), ("invert", C.c_int32)]


class MitRaggedSeg(C.Structure):
    _fields_ = [("pixel_start", C.c_int64), ("group_start", C.c_int64), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("_pad", C.c_int32)]


# every symbol include/mit_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mit_last_error": (C.c_char_p, []),
    "mit_abi_version": (C.c_int, []),
    "mit_source_digest": (C.c_char_p, []),
    "mit_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mit_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "mit_conv_gemm": (C.c_int, [C.POINTER(MitConvGemm), C.c_void_p]),
    "mit_conv_gemm_cfg": (C.c_int, [C.POINTER(MitConvGemm), C.c_int, C.c_void_p]),
    "mit_conv_gemm_config_name": (C.c_char_p, [C.c_int]),
    "mit_conv_gemm_config_kernel": (C.c_char_p, [C.c_int]),
    "mit_gemm_split_pack": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "mit_gemm_mode_set": (C.c_int, [C.c_int]),
    "mit_gemm_mode_get": (C.c_int, []),
    "mit_gemm_split_min_tiles": (C.c_int64, [C.c_int64]),
    "mit_conv_small_cout": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "mit_convnext_mlp_supported": (C.c_int, [C.c_int]),
    "mit_convnext_mlp": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_prof_enable": (C.c_int, [C.c_int]),
    "mit_prof_tag_next": (C.c_int, [C.c_double]),
    "mit_prof_read": (C.c_int, [C.POINTER(MitProfStat), C.c_int, C.POINTER(C.c_int)]),
    "mit_prof_dump": (C.c_int, [C.c_char_p]),
    "mit_prof_kernels_read": (C.c_int, [C.POINTER(MitProfKernelStat), C.c_int, C.POINTER(C.c_int)]),
    "mit_wino43_input": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "mit_wino43_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "mit_ocr_warp_lines": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p]),
    "mit_fft_cols": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                               C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "mit_rfft_rows_supported": (C.c_int, [C.c_int]),
    "mit_rfft_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "mit_irfft_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                 C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_void_p]),
    "mit_bilateral_u8c3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "mit_mask_dilate_jobs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mit_binarize_u8": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_densecrf_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int]),
    "mit_densecrf_refine": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_ctd_refine_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int]),
    "mit_ctd_refine_hist": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p]),
    "mit_ctd_refine_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_void_p]),
    "mit_ctd_refine_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_void_p]),
    "mit_lama_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mit_lama_prep_padded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mit_lama_mpe_index": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "mit_lama_mpe_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "mit_lama_mpe_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mit_ctd_prep": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mit_maxpool_nhwc": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p]),
    "mit_maxpool2d_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    "mit_avgpool2_nhwc": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p]),
    "mit_copy_channels": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "mit_map_to_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p]),
    "mit_axpy": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_boxes_from_bitmap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mit_boxes_debug_stamps": (C.c_int, [C.c_void_p]),
    "mit_boxes_from_bitmap_dev_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mit_boxes_from_bitmap_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                            C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mit_merge_mask_list": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mit_otsu_from_hist": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mit_find_contours_count": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "mit_mask_assign_lines": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "mit_mask_line_crops": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mit_ocr_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mit_dwconv_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p]),
    "mit_cotenant_safe_set": (C.c_int, [C.c_int]),
    "mit_quad_pair_distances": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mit_ocr48_decode_rows_max_set": (C.c_int, [C.c_int]),
    "mit_pgemm": (C.c_int, [C.POINTER(MitPGemm), C.c_void_p]),
    "mit_pgemm_tile_name": (C.c_char_p, [C.c_int]),
    "mit_pgemm_supported": (C.c_int, [C.POINTER(MitPGemm)]),
    "mit_split_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_join_planes": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_dwconv_nhwc_ragged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "mit_dwconv_nhwc_ragged_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mit_layernorm": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                C.c_float, C.c_void_p]),
    "mit_xpos_rotate": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.POINTER(MitXposTables), C.c_void_p]),
    "mit_attention": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p]),
    "mit_attention_lines_xpos_max_len": (C.c_int, [C.c_int]),
    "mit_attention_self_rows_set": (C.c_int, [C.c_int]),
    "mit_attention_lines_xpos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.POINTER(MitXposTables), C.c_void_p]),
    "mit_memory_kv_lines": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int64, C.c_int, C.c_int, C.POINTER(MitXposTables), C.c_void_p]),
    "mit_attention_heads": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    "mit_avgpool_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p]),
    "mit_affine_act_nhwc": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                      C.c_int, C.c_void_p]),
    "mit_u8_to_f32_nhwc4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "mit_sigmoid_inplace": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_gelu_inplace": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "mit_logsoftmax_top5": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mit_ocr48_decode_workspace_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "mit_ocr48_decode": (C.c_int, [C.POINTER(MitOcr48Decoder), C.POINTER(MitOcr48DecodeArgs), C.c_void_p]),
    "mit_resize_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "mit_select_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "mit_lama_post": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_void_p]),
}

_lib = None


def lib_path() -> Path:
    return Path(__file__).resolve().parent / "libmit_hip.so"


def _stale(path: Path, want: str) -> bool:
    """True when the binary at ``path`` was built from other sources than the tree's.  The digest is read from the marker
    string in the file, not through dlopen: a stale library must not stay mapped when its replacement is loaded."""
    needle = b"MIT_SOURCE_DIGEST=" + want.encode()
    try:
        with open(path, "rb") as f:  # chunked scan: the library is ~10 MB and every process start checks it
            tail = b""
            while True:
                chunk = f.read(1 << 20)
                if not chunk:
                    return True
                if needle in tail + chunk:
                    return False
                tail = chunk[-len(needle):]
    except OSError:
        return True


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load ``libmit_hip.so``.  A missing library, or one built from other sources than the tree's (``mit_source_digest()``
    against ``build.source_digest()``), is rebuilt with hipcc when ``build_if_missing`` — otherwise that is an error: old
    kernels are never validated or measured silently.  Raises on any failure."""
    global _lib
    if _lib is not None:
        return _lib
    from . import build as _build

    path = lib_path()
    want = _build.source_digest()
    if not path.exists() or _stale(path, want):
        if not build_if_missing:
            what = "is missing" if not path.exists() else "was built from different sources than this tree"
            raise RuntimeError(f"{path} {what}: run `python -m manga_image_translator_amd.build`")
        # several processes (the ranks of a multi-GPU launch, pytest workers) may find the library stale at the same moment: one of
        # them builds under an exclusive file lock, the others wait and re-check; the build links to a temporary name and renames
        import fcntl

        with open(str(path) + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not path.exists() or _stale(path, want):
                    _build.build(force=True)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mit_abi_version() != MIT_ABI_VERSION:
        raise RuntimeError(f"libmit_hip.so ABI {lib.mit_abi_version()} != binding {MIT_ABI_VERSION}")
    if lib.mit_source_digest().decode() != want:
        raise RuntimeError(f"{path} does not match the sources it was just built from (digest mismatch)")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    """Turn a non-zero status into RuntimeError carrying ``mit_last_error()``."""
    if rc != 0:
        msg = load().mit_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what}: {msg}" if what else msg)
