"""ctypes binding of libs2l_hip.so (the C-ABI declared in include/s2l_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p, POINTER

from .build import LIB

S2L_NUM_TENSORS = 42
S2L_PAD_MAY, S2L_PAD_DEFAULT = 0, 1
S2L_POSE_OBS2CAN, S2L_POSE_CAN2OBS, S2L_POSE_CAN2OBS_INV = 0, 1, 2
S2L_SAMPLE_ZEROS, S2L_SAMPLE_BORDER = 0, 1

# order of the pointer table of s2l_pack_weights == enum S2L_T_* in include/s2l_hip.h
TENSOR_ORDER = [
    "encoder_conv.0.weight", "encoder_conv.0.bias", "encoder_conv.2.weight", "encoder_conv.2.bias",
    "encoder_conv.4.weight", "encoder_conv.4.bias", "encoder_conv.6.weight", "encoder_conv.6.bias",
    "encoder_fc1.0.weight", "encoder_fc1.0.bias", "encoder_fc1.2.weight", "encoder_fc1.2.bias",
    "fc_uv.weight", "fc_uv.bias", "fc_audio.weight", "fc_audio.bias", "fc_time.weight", "fc_time.bias",
    "fc_uv_skip.weight", "fc_uv_skip.bias", "fc_audio_skip.weight", "fc_audio_skip.bias",
    "fc_time_skip.weight", "fc_time_skip.bias",
    "pts_linears.0.weight", "pts_linears.0.bias", "pts_linears.1.weight", "pts_linears.1.bias",
    "pts_linears.2.weight", "pts_linears.2.bias", "pts_linears.3.weight", "pts_linears.3.bias",
    "pts_linears.4.weight", "pts_linears.4.bias", "pts_linears.5.weight", "pts_linears.5.bias",
    "pts_linears.6.weight", "pts_linears.6.bias", "pts_linears.7.weight", "pts_linears.7.bias",
    "output_linear.weight", "output_linear.bias",
]
assert len(TENSOR_ORDER) == S2L_NUM_TENSORS

_ERRORS = {-1: "S2L_E_NULL (null pointer)", -2: "S2L_E_SIZE (bad size)", -3: "S2L_E_ALIGN (pointer not 16-byte aligned)",
           -4: "S2L_E_GEOMETRY (geometry the reference cannot evaluate either, e.g. a lip box entirely outside the face frame)",
           -5: "S2L_E_UNSUPPORTED (a kernel form only libs2l_hip_ref.so holds)"}

EXPORTS = {
    "s2l_version": (c_char_p, []),
    "s2l_packed_floats": (c_int64, []),
    "s2l_pack_weights": (c_int, [POINTER(c_void_p), POINTER(c_float), c_void_p, c_void_p]),
    "s2l_audio_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_frame_vectors": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_frame_front": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_pixel_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_render_lip": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_render16_packed_halves": (c_int64, []),
    "s2l_pack_render16": (c_int, [c_void_p, c_void_p, c_void_p]),
    "s2l_render_lip_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_set_render_cus": (c_int, [c_int]),
    "s2l_rgb_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_embed_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "s2l_composite": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "s2l_composite_stream_work_bytes": (c_int64, [c_int, c_int, c_int, c_int, c_int64]),
    "s2l_composite_stream": (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_int64, c_void_p]),
    "s2l_composite_train": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "s2l_composite_backward_lip": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_saved_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_backward_work_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_forward_saved": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_saved_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_train_work_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_grad_floats": (c_int64, []),
    "s2l_unet_pack_raw": (c_int, [POINTER(c_void_p), c_void_p, c_void_p]),
    "s2l_unet_train_forward": (c_int, [c_void_p, POINTER(c_void_p), c_float, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_backward": (c_int, [c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_forward_bf16": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_backward_bf16": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_frames_saved_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_train_frames_scratch_floats": (c_int64, [c_int64]),
    "s2l_unet_train_frames_work_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_train_forward_frames": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_backward_frames": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_forward_saved_window": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                              c_int64, c_void_p]),
    "s2l_unet_backward_window": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_int64, c_void_p]),
    "s2l_unet_train_frames_h_saved_halves": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_train_frames_h_scratch_floats": (c_int64, [c_int64]),
    "s2l_unet_train_frames_h_work_halves": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_train_forward_frames_h": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                                c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_forward_frames_h_fused": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_float, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                                      c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_backward_frames_h": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                 c_int64, c_void_p]),
    "s2l_convh_layer": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_debug_conv_layer_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                         c_int64, c_void_p]),
    "s2l_set_unet_half_kernel": (c_int, [c_int]),
    "s2l_debug_convh_layer_stats": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_debug_convh_layer_bstats": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_debug_conv_wgrad_h": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_debug_bf16_mfma_rate": (c_int, [c_int64, c_int, c_void_p, c_void_p]),
    "s2l_unet_saved_h_halves": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_backward_h_work_halves": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_forward_saved_h": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_backward_h": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int64,
                                    c_void_p]),
    "s2l_unet_train_backward_frames_grads": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_unet_train_frames_h_work_halves_grads": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_train_backward_frames_h_grads": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_set_unet_conv_kernel": (c_int, [c_int]),
    "s2l_set_unet_split_kernel": (c_int, [c_int]),
    "s2l_set_render_shape": (c_int, [c_int]),
    "s2l_set_rows_kernel": (c_int, [c_int]),
    "s2l_train_backward_bf16_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_set_bf16_forward_kernel": (c_int, [c_int]),
    "s2l_unet_packed16_halves": (c_int64, []),
    "s2l_unet_pack16": (c_int, [c_void_p, ctypes.c_float, c_void_p, c_void_p]),
    "s2l_unet_packed16x3_halves": (c_int64, []),
    "s2l_unet_pack16x3": (c_int, [c_void_p, ctypes.c_float, c_void_p, c_void_p]),
    "s2l_unet_forward_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_crop_resize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "s2l_crop_resize_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int64,
                                         c_void_p]),
    "s2l_predict_lip_image_work_floats": (c_int64, [c_int64]),
    "s2l_predict_lip_image": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p,
                                      c_int64, c_void_p]),
    "s2l_ensemble_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_int64,
                                  c_void_p]),
    "s2l_ensemble_rows_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64,
                                        c_int64, c_void_p]),
    "s2l_segment_colsums": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "s2l_unfold_first_layer": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 13),
    "s2l_ensemble_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_ensemble_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_ensemble_reduce_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_ensemble_backward_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_split_work_floats": (c_int64, [c_int64]),
    "s2l_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_small_outer": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_audio_grad_floats": (c_int64, []),
    "s2l_audio_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_mse": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_adam_chunk": (c_int64, []),
    "s2l_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double, c_double, c_double, c_double, c_double, c_int64, c_void_p, c_void_p]),
    "s2l_unet_packed_floats": (c_int64, []),
    "s2l_unet_work_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_unet_pack": (c_int, [POINTER(c_void_p), c_float, c_void_p, c_void_p]),
    "s2l_unet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_composite_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "s2l_rel_pose": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "s2l_warp_grid": (c_int, [c_void_p, c_int64, c_void_p, c_float, c_int, c_void_p, c_void_p, c_int, c_int, c_int64,
                              c_void_p]),
    "s2l_grid_sample": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64,
                                c_void_p]),
    "s2l_depth_photo_work_floats": (c_int64, [c_int, c_int]),
    "s2l_depth_photo_loss": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_float, c_void_p,
                                     c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_lpips_packed_floats": (c_int64, []),
    "s2l_lpips_work_floats": (c_int64, [c_int, c_int, c_int64]),
    "s2l_lpips_pack": (c_int, [c_void_p, c_void_p, c_void_p]),
    "s2l_lpips_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_lpips_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_lpips_forward_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_lpips_backward_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "s2l_syncnet_packed_floats": (c_int64, []),
    "s2l_syncnet_work_floats": (c_int64, [c_int64]),
    "s2l_syncnet_pack": (c_int, [POINTER(c_void_p), c_float, c_void_p, c_void_p]),
    "s2l_syncnet_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_sync_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "s2l_syncnet_face_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_syncnet_forward_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_syncnet_face_backward_prefix": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_syncnet_forward_pair_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_syncnet_face_backward_prefix_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "s2l_sync_window": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "s2l_sync_window_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_void_p]),
    "s2l_bf16_packed_halves": (c_int64, []),
    "s2l_bf16_rows_padded": (c_int64, [c_int64]),
    "s2l_pack_bf16": (c_int, [POINTER(c_void_p), c_void_p, c_void_p, c_void_p]),
    "s2l_ensemble_rows_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64,
                                       c_int64, c_void_p]),
    "s2l_train_forward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_train_backward_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_wgrad_bf16_work_floats": (c_int64, []),
    "s2l_wgrad_bf16": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_rows_to_tiles_bf16": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "s2l_out_grad_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_to8b": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "s2l_from8b": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None


class S2LError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libs2l_hip.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        path = os.environ.get("S2L_LIB", LIB)   # A/B builds of the same ABI (kernel experiments)
        if not os.path.exists(path):
            raise S2LError(f"{path} not found: build it with `python -m speech2lip_amd.build` "
                           "(there is no CPU fallback for the lip-render path)")
        lib = ctypes.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


_ref_lib = None


def load_reference() -> ctypes.CDLL:
    """libs2l_hip_ref.so: the same ABI built with -DS2L_WITH_REFERENCE_KERNELS, i.e. WITH the non-default kernel forms that exist only
    to pin the default ones bit for bit (the four-wave and the alternating-roles half-width convolutions, the generated-assembly
    split convolution).  TEST INFRASTRUCTURE: only tests/ and tools/ may call this; nothing under speech2lip_amd/ does."""
    global _ref_lib
    if _ref_lib is None:
        path = os.path.join(os.path.dirname(LIB), "libs2l_hip_ref.so")
        if not os.path.exists(path):
            raise S2LError(f"{path} not found: build it with `python -m speech2lip_amd.build --ref`")
        lib = ctypes.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _ref_lib = lib
    return _ref_lib


class reference_kernels:
    """`with _abi.reference_kernels() as ref:` -- inside the block every call of the package goes to libs2l_hip_ref.so (selectors
    set on `ref` take effect there); on exit the product library is back.  Tests and tools only."""

    def __enter__(self):
        global _lib
        load()
        self.saved = _lib
        _lib = load_reference()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise S2LError(f"{what}: {_ERRORS.get(rc, rc)}")
    raise S2LError(f"{what}: hipError_t {rc}")
