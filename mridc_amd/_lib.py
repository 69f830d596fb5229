"""ctypes binding of libmridc_amd.so (C ABI in include/mridc_amd.h).

There is NO CPU fallback: if the shared library is missing or a tensor is not on the GPU the call
raises.  Tensors are passed as raw device pointers plus the current torch HIP stream; torch is used
only to own memory and streams.
"""
import ctypes
import os
import re

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRIDC_AMD_LIB") or os.path.join(_PKG, "lib", "libmridc_amd.so")   # MRIDC_AMD_LIB: another build of the same library (A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "mridc_amd.h")

NORM = {"backward": 0, "ortho": 1, "forward": 2, "none": 3}
MASK_U8, MASK_F32 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
PAD_ZERO, PAD_REPLICATE = 0, 1

_p, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

_SIGNATURES = {
    "mrx_version": ([], _i),
    "mrx_last_error": ([], ctypes.c_char_p),
    "mrx_stream_capture_id": ([_p], _i64),
    "mrx_arith": ([], _i),
    "mrx_checks_enabled": ([], _i),
    "mrx_poisson_disc_mask": ([_i, _i, _i, _p, _p, ctypes.c_double, ctypes.c_double, ctypes.c_uint64, _p], _i64),
    "mrx_fft_prepare": ([_i, _i], _i),
    "mrx_fft_max_len": ([], _i),
    "mrx_fft2": ([_p, _p, _i64, _i, _i, _i, _i, _i, _p], _i),
    "mrx_roll": ([_p, _p, _i, _i, _p, _p, _p], _i),
    "mrx_complex_mul": ([_p, _p, _p, _i, _p, _p, _p, _i, _p], _i),
    "mrx_complex_conj": ([_p, _p, _i64, _p], _i),
    "mrx_complex_abs": ([_p, _p, _i64, _i, _p], _i),
    "mrx_rss": ([_p, _p, _i64, _i64, _i64, _p], _i),
    "mrx_rss_complex": ([_p, _p, _i64, _i64, _i64, _p], _i),
    "mrx_sense": ([_p, _p, _p, _i64, _i64, _i64, _p], _i),
    "mrx_apply_mask": ([_p, _p, _p, _i, _i, _i, _i, _p, _p], _i),
    "mrx_sens_expand": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_sens_reduce": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_llg": ([_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_fft_cols": ([_p, _p, _i64, _i, _i, _i, _i, _i, _p], _i),
    "mrx_llg_hinv_work_floats": ([_i, _i, _i, _i], _i64),
    "mrx_llg_hinv": ([_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_soft_dc": ([_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_coil_sum": ([_p, _p, _i, _p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_dc_bcast": ([_p, _i, _p, _p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _p], _i),
    "mrx_lincomb": ([_p, _i64, _p, _i64, _p, _i, _p, _i64, _p], _i),
    "mrx_cdot_work_floats": ([_i], _i64),
    "mrx_cdot": ([_p, _p, _p, _p, _i, _i64, _p], _i),
    "mrx_cg_step": ([_p, _p, _p, _p, _p, _p, _i, _i64, _p], _i),
    "mrx_cg_dir": ([_p, _p, _p, _p, _i, _i64, _p], _i),
    "mrx_conv2dgru_supported": ([_i, _i, _i], _i),
    "mrx_conv2dgru_pack_floats": ([_i], _i64),
    "mrx_conv2dgru_pack": ([_p, _p, _p, _p, _i, _p], _i),
    "mrx_conv2dgru_cell_1x1": ([_p, _p, _p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_conv2dgru_cell_1x1_xmax": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_mul_sigmoid": ([_p, _p, _p, _i64, _p], _i),
    "mrx_gru_blend": ([_p, _p, _p, _p, _p, _i64, _p], _i),
    "mrx_conv3x3_wino_supported": ([_i, _i, _i, _i], _i),
    "mrx_conv3x3_wino": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_conv_to_complex": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_conv_wgrad_work_floats": ([_i, _i, _i, _i, _i, _i], _i64),
    "mrx_conv_wgrad": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_reppad_fold": ([_p, _p, _i64, _i, _i, _i, _p], _i),
    "mrx_relu_bwd_work_floats": ([_i], _i64),
    "mrx_relu_bwd": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_conv_bf16_supported": ([_i, _i, _i, _i], _i),
    "mrx_conv_bf16_pack_bytes": ([_i, _i, _i], _i64),
    "mrx_conv_bf16_pack": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_conv2d_bf16": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_conv2d_bf16_ext": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_conv2d_bf16_dgrad_rep": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_reppad_fold_edges": ([_p, _p, _i64, _i, _i, _i, _p], _i),
    "mrx_conv_wgrad_bf16_supported": ([_i, _i, _i, _i], _i),
    "mrx_conv_wgrad_bf16_work_floats": ([_i, _i, _i, _i], _i64),
    "mrx_conv_wgrad_bf16": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_tl_pack_bytes": ([], _i64),
    "mrx_tl_pack": ([_p, _p, _p, _p], _i),
    "mrx_tl_layer_fwd": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_tl_dgrad": ([_p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_tl_dgrad_l2w": ([_p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_tl_fold_edges": ([_p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_tl_cell_part_floats": ([_i, _i, _i], _i64),
    "mrx_tl_cell_bwd": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_tl_cell_reduce": ([_p, _i, _i, _i, _p, _p, _p, _p, _p], _i),
    "mrx_tl_final_gather_max_count": ([_i, _i, _i], _i64),
    "mrx_tl_final_gather_max": ([_p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_absl1_loss_mp": ([_p, _p, _p, _i, _p, _p, _p, _i64, _p], _i),
    "mrx_absl1_loss_bwd_eta": ([_p, _p, _p, _p, _p, _f, _p, _p, _p, _i, _i64, _p], _i),
    "mrx_eta_grad_out_parts": ([_p, _p, _p, _i, _f, _p, _i, _i64, _p], _i),
    "mrx_tl_final_gather": ([_p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_tl_pairs_to_f32": ([_p, _p, _i64, _i64, _p], _i),
    "mrx_tl_f32_to_pairs": ([_p, _p, _i64, _i64, _p], _i),
    "mrx_tl_wgrad_in_work_floats": ([_i, _i, _i, _i], _i64),
    "mrx_tl_wgrad_in": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_conv_wgrad_bf16_pairs": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_conv_wgrad_bf16_xcb": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_act_bwd": ([_p, _p, _p, _i64, _i, _f, _p], _i),
    "mrx_inorm_act_bwd_work_floats": ([_i64, _i64], _i64),
    "mrx_inorm_act_bwd": ([_p, _p, _p, _p, _p, _i64, _i64, _f, _i, _f, _p], _i),
    "mrx_avgpool2x2_bwd": ([_p, _p, _i64, _i, _i, _p], _i),
    "mrx_cmul_bcast": ([_p, _p, _p, _i64, _i64, _i64, _i, _f, _p], _i),
    "mrx_sens_expand_bwd_pw": ([_p, _p, _p, _p, _p, _i64, _i64, _i64, _f, _p], _i),
    "mrx_dc_combine_bwd_work_doubles": ([], _i64),
    "mrx_dc_combine_bwd": ([_p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_pixel_unshuffle2": ([_p, _p, _i64, _i, _i, _p], _i),
    "mrx_relu_bwd_acc": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_eta_grad_in": ([_p, _p, _p, _p, _i, _i64, _p], _i),
    "mrx_g4_to_complex": ([_p, _p, _i, _i64, _p], _i),
    "mrx_eta_grad_out": ([_p, _p, _p, _p, _i, _i64, _p], _i),
    "mrx_conv_wgrad_bf16_any_work_floats": ([_i, _i, _i, _i, _i, _i], _i64),
    "mrx_conv_wgrad_bf16_any": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_absl1_work_floats": ([], _i64),
    "mrx_absl1_loss": ([_p, _p, _p, _p, _p, _i64, _p], _i),
    "mrx_absl1_loss_bwd": ([_p, _p, _p, _p, _p, _f, _p, _i64, _p], _i),
    "mrx_adam_step": ([_p, _p, _p, _p, _i64, _f, _f, _f, _f, _i, _f, _p], _i),
    "mrx_llg_hinv_parts": ([_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_llg372_supported": ([_i], _i),
    "mrx_llg372_operand_floats": ([_i, _i, _i], _i64),
    "mrx_llg372_work_floats": ([_i, _i, _i], _i64),
    "mrx_llg372_gather": ([_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_llg372_gather_q": ([_p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_llg372_const_plane": ([_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_llg372_prepare": ([_p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_llg372": ([_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_pfa372_prepare_maps": ([_p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_pfa372_expand": ([_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_pfa372_expand_reduce": ([_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_pfa372_reduce": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_llg_cols_dc": ([_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_tile4_cols": ([_p, _p, _i64, _i, _i, _p], _i),
    "mrx_pfa372_expand_t4": ([_p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_pfa372_expand_t4_gather": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_llg_cols_dc_t4": ([_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_llg_cols_dc_t4_supported": ([_i, _i], _i),
    "mrx_pfa372_reduce_t4": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _i, _p], _i),
    "mrx_rim_layer_indrnn_packed_llg": ([_p, _p, _i, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_conv1x1_64_pack": ([_p, _p, _p], _i),
    "mrx_conv1x1_64": ([_p, _p, _p, _p, _p, _p, _i, _i64, _i, _f, _p], _i),
    "mrx_sens_expand_rows": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_sens_reduce_rows": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_sens_expand_rows_dc": ([_p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_conv1x1_sq_supported": ([_i, _i], _i),
    "mrx_conv1x1_sq_pack_floats": ([_i], _i64),
    "mrx_conv1x1_sq_head128": ([_p, _p, _p, _i, _i64, _p], _i),
    "mrx_conv1x1_sq_pack": ([_p, _p, _i, _p], _i),
    "mrx_conv1x1_sq": ([_p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _f, _p], _i),
    "mrx_conv1x1_sq_xmax_supported": ([_i], _i),
    "mrx_conv1x1_sq_xmax": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _f, _p], _i),
    "mrx_conv1x1_sq_p16": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _f, _p], _i),
    "mrx_concat_channels": ([_p, _p, _p, _i, _i, _i, _i64, _p], _i),
    "mrx_unet_conv3x3_work_floats": ([_i, _i, _i, _i], _i64),
    "mrx_unet_conv3x3_pack_floats": ([_i, _i], _i64),
    "mrx_unet_conv3x3_pack": ([_p, _i, _i, _p, _p], _i),
    "mrx_unet_conv3x3_h": ([_p, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p], _i),
    "mrx_unet_conv3x3_p16": ([_p, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p], _i),
    "mrx_unet_conv3x3_hc_ticket_ints": ([_i, _i], _i64),
    "mrx_unet_conv3x3_hc": ([_p, _p, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p], _i),
    "mrx_conv3x3_h_supported": ([_i, _i, _i, _i], _i),
    "mrx_conv3x3_h": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_conv3x3_p16": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_unet_conv3x3": ([_p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p], _i),
    "mrx_unet_conv_transpose2x2_work_floats": ([_i, _i, _i, _i], _i64),
    "mrx_unet_conv_transpose2x2": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p], _i),
    "mrx_unet_avgpool": ([_p, _p, _p, _i64, _i, _i, _f, _p], _i),
    "mrx_unet_apply": ([_p, _p, _p, _i64, _i64, _f, _p], _i),
    "mrx_unet_conv1x1": ([_p, _p, _p, _p, _p, _i, _i, _i, _i64, _f, _p], _i),
    "mrx_unet_cnorm_work_floats": ([_i], _i64),
    "mrx_unet_cnorm_pad": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_unet_conv1x1_cunnorm": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_hard_dc": ([_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_vs_average": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_dc_combine": ([_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p], _i),
    "mrx_conv2d": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_rim_layer_indrnn": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_rim_layer_pack_floats": ([_i, _i, _i], _i64),
    "mrx_rim_layer_pack": ([_p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_rim_layer_supported": ([_i, _i, _i, _i], _i),
    "mrx_rim_layer_indrnn_packed": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_div_rss_complex": ([_p, _p, _i64, _i64, _i64, _p], _i),
    "mrx_max_abs_work_floats": ([], _i64),
    "mrx_max_abs": ([_p, _i64, _i, _p, _p, _p], _i),
    "mrx_div_by_device_scalar": ([_p, _p, _p, _i64, _i, _p], _i),
    "mrx_recon_metrics_work_floats": ([], _i64),
    "mrx_recon_metrics": ([_p, _p, _p, _p, _i64, _p], _i),
    "mrx_rim_layer2_sb_pack_floats": ([], _i64),
    "mrx_rim_layer1_xmax_supported": ([_i, _i, _i, _i], _i),
    "mrx_rim_layer_indrnn_packed_xmax": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_rim_layer_indrnn_packed_llg_xmax": ([_p, _p, _i, _f, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_rim_layer2_f16_pack_floats": ([], _i64),
    "mrx_rim_layer2_f16_pack": ([_p, _p, _p, _p, _p], _i),
    "mrx_rim_layer2_f16": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_rim_layer2_sb_pack": ([_p, _p, _p, _p, _p], _i),
    "mrx_conv3x3_sb_supported": ([_i, _i, _i, _i], _i),
    "mrx_conv_sbs_supported": ([_i, _i, _i, _i], _i),
    "mrx_conv_sbs_pack_floats": ([_i, _i], _i64),
    "mrx_conv_sbs_pack": ([_p, _p, _i, _i, _i, _p], _i),
    "mrx_conv_sbs": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_conv_sbs_p16": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_conv3x3_sb": ([_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_conv3x3_sb_chain": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p], _i),
    "mrx_rim_layer2_sb_taps": ([_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_rim_final_gather": ([_p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_cb8_convert": ([_p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_rim_layer1_cb8": ([_p, _i, _p, _p, _i, _f, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_rim_layer2_f16_cb8": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_rim_taps_q_edge_floats": ([_i, _i, _i], _i64),
    "mrx_rim_layer2_f16_cb8_q": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_amp16_pack_floats": ([_i], _i64),
    "mrx_amp16_layer1_pack": ([_p, _p, _p, _i, _p], _i),
    "mrx_amp16_layer2_pack": ([_p, _p, _p, _p, _p], _i),
    "mrx_amp16_layer1": ([_p, _i, _p, _p, _i, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_amp16_layer2": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_rim_final_gather_q": ([_p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_cnorm_work_doubles": ([_i], _i64),
    "mrx_cnorm_stats": ([_p, _i, _i64, ctypes.c_double, _i, _p, _p, _p], _i),
    "mrx_cnorm_apply": ([_p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_cnorm_unapply": ([_p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_taps_gather": ([_p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_rim_layer2_sb": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_rim_layer_wino_pack_floats": ([_i, _i], _i64),
    "mrx_rim_layer_wino_pack": ([_p, _p, _p, _i, _i, _p], _i),
    "mrx_rim_layer_indrnn_wino": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_indrnn_cell": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_rim_final": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_gru_gates": ([_p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_mgu_gates": ([_p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_gru_gates_bwd": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_mgu_gates_bwd": ([_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_gated_cell_supported": ([_i, _i, _i, _i], _i),
    "mrx_gated_cell_pack_floats": ([_i, _i, _i], _i64),
    "mrx_gated_cell_pack": ([_p, _p, _p, _i, _i, _i, _p], _i),
    "mrx_gated_cell_1x1": ([_p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _p], _i),
    "mrx_gated_cell_1x1_xmax": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _i, _p], _i),
    "mrx_norm_work_floats": ([_i64, _i64], _i64),
    "mrx_instance_norm_act": ([_p, _p, _p, _i64, _i64, _f, _i, _f, _p], _i),
    "mrx_conv2d_stats_work_floats": ([_i, _i, _i, _i], _i64),
    "mrx_conv2d_stats_supported": ([_i, _i, _i, _i, _i, _i], _i),
    "mrx_conv2d_stats": ([_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_instance_norm_apply": ([_p, _p, _p, _i64, _i64, _f, _i, _f, _p], _i),
    "mrx_instance_norm_apply_tiles": ([_p, _p, _p, _i, _i, _i, _i, _f, _i, _f, _p], _i),
    "mrx_group_norm_stats": ([_p, _p, _p, _p, _i64, _i64, _p], _i),
    "mrx_group_norm_apply": ([_p, _p, _p, _p, _i64, _i64, _i, _p], _i),
    "mrx_group_norm_bwd": ([_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i, _p], _i),
    "mrx_pad2d": ([_p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_avg_pool2x2": ([_p, _p, _i64, _i, _i, _p], _i),
    "mrx_conv_transpose2x2_stats_work_floats": ([_i, _i, _i, _i], _i64),
    "mrx_conv_transpose2x2_stats": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_conv_transpose2x2": ([_p, _p, _p, _i, _i, _i, _i, _i, _p], _i),
    "mrx_dc_residual": ([_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p], _i),
    "mrx_qmri_signal": ([_p, _p, _p, _p, _p, _i, _p, _i64, _i64, _f, _p], _i),
    "mrx_qmri_grad": ([_p, _p, _p, _p, _p, _p, _i, _p, _i64, _i64, _f, _f, _p], _i),
    "mrx_scale": ([_p, _p, _i64, _f, _i, _p], _i),
    "mrx_qrim_update": ([_p, _p, _p, _i, _i, _i64, _p], _i),
    "mrx_ssim_work_floats": ([_i, _i, _i], _i64),
    "mrx_ssim_loss": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p], _i),
    "mrx_copy_channels": ([_p, _p, _i, _i, _i64, _i, _i, _p], _i),
}

_lib = None


ARITH_NAMES = ("f16x2", "bf16x3", "fp32")


def arith():
    """The arithmetic route (environment MRIDC_AMD_ARITH, see include/mridc_amd.h): "f16x2" (default), "bf16x3" or "fp32" -- read on the
    Python side exactly as libmridc_amd's mrx_arith reads it."""
    v = os.environ.get("MRIDC_AMD_ARITH", "f16x2")
    return v if v in ARITH_NAMES else "f16x2"


def precision():
    """The inference precision of the regularisers (environment MRIDC_AMD_PRECISION): 32 (default: fp32-class results on every route of `arith()`) or
    16 -- the reference's `trainer.precision: 16` (base_cirim_run.yaml:132, base_vn_run.yaml:98): fp16 operands and hidden states in the two RIM layers
    (csrc/rim_amp16.hip), one-term fp16 operands in the U-Net's 3x3 convolutions (mrx_unet_conv3x3_p16); FFT / data consistency / eta in fp32.  A model
    attribute (RIMBlock.precision set by CIRIM, VarNet.precision, UNet.precision: from the trainer / cfg) overrides it."""
    return 16 if os.environ.get("MRIDC_AMD_PRECISION", "32").strip().lower() in ("16", "fp16", "16-mixed", "amp16") else 32


def declared_symbols():
    """Function names declared in include/mridc_amd.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrx_[a-z0-9_]+)\s*\(", text)))


def written_pointer_args():
    """{function name: positions of its non-const pointer parameters other than `stream`} read off include/mridc_amd.h: the parameters a call may
    WRITE through.  The header is const-correct (tests/test_host_logic.py checks that every declared function parses), so this list needs no
    maintenance when an entry point is added."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(mrx_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        pos = []
        for i, a in enumerate(args.split(",")):
            a = a.strip()
            if "*" in a and not a.startswith("const") and not re.search(r"\bstream$", a):
                pos.append(i)
        out[name] = tuple(pos)
    return out


# ---- operand bounds: who owns them -------------------------------------------------------------------------------------------------------------
# A two-term fp16 kernel scales its input by a device scalar >= max |x|.  Producers that know that maximum (a kernel that folds it out of its own
# accumulators, an instance norm whose outputs are bounded analytically) leave it here, consumers ask for it -- a table of MEMORY RANGES owned by
# this binding, not an attribute on tensor objects:
#   * an entry is (address range, the tensor object it was attached to (weak), that tensor's version, the bound scalar);
#   * every call into the library invalidates the entries overlapping a range it may write: `ptr()` hands the call the tensor's address range, and
#     the wrapper installed on each entry point looks at its non-const pointer parameters (written_pointer_args).  That covers `out=` tensors,
#     VIEWS of a bounded tensor and in-place kernels alike -- the library writes through raw pointers and never bumps a tensor version;
#   * writes by torch itself bump the version (shared by all views), a freed tensor kills the weak reference: both fail the look-up.
# Round 4 kept the bound as `tensor._mrx_bound` and relied on each op remembering to drop it from its `out` argument.
class _Ptr(ctypes.c_void_p):
    """c_void_p of a tensor's first byte that also knows the byte range [lo, hi) the tensor covers."""
    __slots__ = ("lo", "hi")


_BOUNDS = {}            # lo -> (hi, weakref(tensor), version, bound scalar tensor)


def bound_attach(t, bound):
    """Remember `bound` (1-element device tensor, >= max |t| once the stream has run the producer) for tensor t."""
    import weakref
    if len(_BOUNDS) >= 64:
        for k in [k for k, e in _BOUNDS.items() if e[1]() is None]:
            del _BOUNDS[k]
        while len(_BOUNDS) >= 64:
            _BOUNDS.pop(next(iter(_BOUNDS)))
    lo, hi = _span(t)
    _BOUNDS[lo] = (hi, weakref.ref(t), t._version, bound)
    return t


def bound_of(t):
    """The bound attached to exactly this tensor object, if nothing has written its memory since; else None."""
    e = _BOUNDS.get(t.data_ptr()) if _BOUNDS else None
    if e is None or e[1]() is not t or e[2] != t._version or _span(t)[1] != e[0]:
        return None
    return e[3]


def bound_note_write(lo, hi):
    for k in [k for k, e in _BOUNDS.items() if k < hi and e[0] > lo]:
        del _BOUNDS[k]


def _span(t):
    lo = t.data_ptr()
    if t.is_contiguous():
        return lo, lo + t.numel() * t.element_size()
    ext = 1 + sum((int(n) - 1) * int(st) for n, st in zip(t.shape, t.stride())) if t.numel() else 0
    return lo, lo + ext * t.element_size()


def _tracked(fn, positions):
    def call(*args):
        if _BOUNDS:
            for i in positions:
                p = args[i]
                if type(p) is _Ptr:
                    bound_note_write(p.lo, p.hi)
        return fn(*args)
    call.__name__ = getattr(fn, "__name__", "mrx_call")
    call.raw = fn
    return call


def lib():
    """Load libmridc_amd.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m mridc_amd._build` (hipcc, gfx950). "
                "mridc_amd has no CPU or PyTorch fallback path.")
        L = ctypes.CDLL(LIB_PATH)
        writes = written_pointer_args()
        # a declaration the header parser does not match would leave its entry point silently UNTRACKED (a stale operand bound could then be
        # used): every bound entry point must be found in the header, with as many parameters as its ctypes signature
        missing = [n for n, (args, _) in _SIGNATURES.items() if n not in writes or max(writes[n], default=-1) >= len(args)]
        if missing:
            raise RuntimeError(f"{HEADER_PATH}: no parsable declaration for {missing[:5]} (of {len(missing)}): the write-tracking table of "
                               "mridc_amd._lib would be incomplete")
        for name, (args, res) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.argtypes = args
            fn.restype = res
            if writes.get(name):
                setattr(L, name, _tracked(fn, writes[name]))
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mrx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libmridc_amd {what} failed ({rc}): {msg}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "mridc_amd operators run on the MI355X HIP path only: got a tensor on "
                f"'{t.device}'. There is no CPU fallback; move inputs to the GPU (tensor.cuda()).")


def f32c(t):
    """Contiguous fp32 device tensor (the reference calls .contiguous()/.float() the same way at its boundaries)."""
    require_gpu(t)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def ptr(t):
    """Device pointer of t (NULL for None) carrying t's byte range, so that a call that writes through it invalidates the bounds kept for it."""
    if t is None:
        return ctypes.c_void_p(0)
    p = _Ptr(t.data_ptr())
    p.lo, p.hi = _span(t)
    return p


def i64_array(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def mask_args(mask, B, C, H, W):
    """Mask tensor broadcastable to [B,C,H,W,1] -> (contiguous tensor, kind, element strides over (b,c,h,w))."""
    require_gpu(mask)
    m = mask
    if m.dim() == 5 and m.shape[-1] == 1:
        m = m[..., 0]
    while m.dim() < 4:
        m = m.unsqueeze(0)
    if m.dim() != 4:
        raise ValueError(f"mask of shape {tuple(mask.shape)} is not broadcastable to [B,C,H,W,1]")
    for d, n in zip(m.shape, (B, C, H, W)):
        if d not in (1, n):
            raise ValueError(f"mask of shape {tuple(mask.shape)} is not broadcastable to {(B, C, H, W, 1)}")
    if m.dtype in (torch.bool, torch.uint8):
        kind = MASK_U8
        m = m.contiguous()
        if m.dtype == torch.bool:
            m = m.view(torch.uint8)
    else:
        kind = MASK_F32
        m = m.float().contiguous()
    st = [m.stride(i) if m.shape[i] != 1 else 0 for i in range(4)]
    return m, kind, i64_array(st)
