"""ctypes binding of libinsmos_hip.so (include/insmos_hip.h).  There is NO fallback: if the HIP library
is missing or a call fails, this raises -- the product path never routes through oracle/ or any CPU
implementation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libinsmos_hip.so")

c_vp, c_i64, c_int, c_f32, c_sz, c_u32 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_size_t, ctypes.c_uint)
c_f64 = ctypes.c_double

# name -> (restype, argtypes); mirrors include/insmos_hip.h exactly
SIGNATURES = {
    "insmos_version": (c_int, []),
    "insmos_last_hip_error": (c_int, []),
    "insmos_prof_enable": (c_int, [c_int]),
    "insmos_prof_reset": (c_int, []),
    "insmos_prof_read": (c_int, [c_int, c_vp, c_vp, c_vp]),
    "insmos_prof_name": (ctypes.c_char_p, [c_int]),
    "insmos_prof_read_spans": (c_int, [c_int, c_int, c_vp, c_vp]),
    "insmos_prof_read_union": (c_int, [c_int, c_vp, c_vp, c_vp]),
    "insmos_quantize4d_ws_bytes": (c_sz, [c_i64]),
    "insmos_quantize4d": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_quantize4d_ex": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "insmos_level_down4d_ws_bytes": (c_sz, [c_i64]),
    "insmos_level_down4d": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_level_down4d_chain": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_nbr_from_coarse": (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "insmos_nbr81_from_coarse": (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_nbr81_from_coarse_rows": (c_int, [c_vp, c_i64, c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_nbr81_from_coarse_rows_sparse": (c_int, [c_vp, c_i64, c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_const_conv125_from_coarse": (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                 c_int, c_int, c_vp]),
    "insmos_const_conv125_cubes": (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_int, c_int, c_vp, c_vp]),
    "insmos_nbr81_from_coarse_rows_masked": (c_int, [c_vp, c_i64, c_i64, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp,
                                                     c_int, c_vp]),
    "insmos_nbr_down_up": (c_int, [c_vp, c_i64, c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_build_nbr": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_voxelize_mean_ws_bytes": (c_sz, [c_i64]),
    "insmos_voxelize_mean": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp,
                                     c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_down_coords3d_ws_bytes": (c_sz, [c_vp]),
    "insmos_down_coords3d": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_packed_weight_floats": (c_sz, [c_int, c_int, c_int]),
    "insmos_pack_weights_host": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "insmos_sparse_conv": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_vp, c_vp, c_vp, c_int, c_int, c_vp,
                                   c_int, c_int, c_int, c_int, c_vp]),
    "insmos_sparse_conv_rows": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_int, c_int,
                                        c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "insmos_tslice_starts": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp]),
    "insmos_tapc_blocks": (c_sz, [c_i64]),
    "insmos_tapc_words": (c_sz, [c_int, c_i64, c_int]),
    "insmos_tapc_build": (c_int, [c_vp, c_int, c_i64, c_i64, c_int, c_vp, c_vp, c_vp]),
    "insmos_tapc_build_masked": (c_int, [c_vp, c_vp, c_int, c_i64, c_i64, c_int, c_vp, c_vp, c_vp]),
    "insmos_conv_tap_classes": (c_int, [c_int, c_int, c_int, c_int]),
    "insmos_sparse_conv_tapc_rows": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_int, c_i64, c_i64, c_vp, c_vp, c_vp, c_int,
                                             c_int, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "insmos_quantize4d_windows": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "insmos_build_current_points_windows": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_int, c_vp]),
    "insmos_build_current_points_part": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_int, c_int, c_vp]),
    "insmos_voxelize_mean_windows": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp,
                                             c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_voxelize_windows_phased": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_vp,
                                             c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_int, c_vp]),
    "insmos_down_coords3d_ws_bytes_b": (c_sz, [c_vp, c_int]),
    "insmos_down_coords3d_b": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_dense_nbr2d_b": (c_int, [c_int, c_int, c_int, c_vp, c_vp]),
    "insmos_sparse_to_bev_b": (c_int, [c_vp, c_int, c_int, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "insmos_center_decode_select_b": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32,
                                              c_f32, c_f32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_nms_ws_bytes_b": (c_sz, [c_int, c_int]),
    "insmos_nms_rotated_bev_b": (c_int, [c_vp, c_vp, c_int, c_f32, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_gather_preds_b": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "insmos_boxes_to_onehot_scratch_ints_b": (c_sz, [c_int, c_int, c_i64]),
    "insmos_boxes_to_onehot_b": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_f32, c_f32, c_vp, c_i64, c_int, c_int,
                                         c_int, c_vp, c_int, c_vp, c_vp]),
    "insmos_boxes_to_onehot_rows": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_i64, c_int,
                                            c_int, c_int, c_vp, c_int, c_vp, c_vp]),
    "insmos_debug_table_limit": (c_int, [c_i64]),
    "insmos_forward_streams": (c_int, [c_int]),
    "insmos_forward_thread_release": (c_int, []),
    "insmos_rankmap_words": (c_sz, [c_vp, c_int]),
    "insmos_rankmap_ws_bytes": (c_sz, [c_vp, c_int]),
    "insmos_rankmap_from_keys": (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_down_coords3d_rank": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_build_nbr_rank": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_build_nbr_rank_sparse": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_build_nbr_rank_multi": (c_int, [c_vp, c_int, c_int, c_vp]),
    "insmos_forward_regroup": (c_int, [c_int]),
    "insmos_bev_cosplit": (c_int, [c_int]),
    "insmos_bev_skip_ws_bytes": (c_sz, [c_int, c_int, c_int]),
    "insmos_bev_conv3x3_skip_ws": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp,
                                           c_vp, c_sz, c_vp]),
    "insmos_forward_host_marks": (c_int, [c_vp, c_sz]),
    "insmos_regroup_rows3d": (c_int, [c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_regroup_ws_bytes": (c_sz, [c_i64]),
    "insmos_regroup_rows3d_global": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_regroup_apply_voxels": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "insmos_forward_windows": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_sz, c_vp, c_vp]),
    "insmos_tslice_starts_batched": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    "insmos_bev_conv3x3": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "insmos_bev_conv3x3_skip": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp,
                                        c_vp]),
    "insmos_bev_distance_map_ws_bytes": (c_sz, [c_int, c_int, c_int]),
    "insmos_bev_distance_map": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp, c_vp, c_sz, c_vp]),
    "insmos_bev_skip_executed_pairs": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "insmos_bev_constant_ws_floats": (c_sz, [c_int, c_int]),
    "insmos_bev_constant": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "insmos_deconv_head": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
    "insmos_deconv_head_skip": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int,
                                        c_int, c_vp, c_vp]),
    "insmos_deconv_head_constant_ws_floats": (c_sz, [c_int]),
    "insmos_deconv_head_constant": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "insmos_deconv_head_skip_active_sites": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    "insmos_debug_conv_force": (c_int, [c_int, c_int, c_int]),
    "insmos_debug_conv_quad": (c_int, [c_int]),
    "insmos_debug_conv_split_half": (c_int, [c_int, c_int]),
    "insmos_debug_conv_row32": (c_int, [c_int]),
    "insmos_debug_conv_wide": (c_int, [c_int]),
    "insmos_debug_conv_rowlane": (c_int, [c_int, c_int]),
    "insmos_debug_conv_lds": (c_int, [c_int]),
    "insmos_debug_conv_lds_stats": (c_int, [c_vp, c_int]),
    "insmos_debug_dw_kernel": (c_int, [c_int]),
    "insmos_conv_precision": (c_int, [c_int]),
    "insmos_conv_precision_thread": (c_int, [c_int]),
    "insmos_split_weights_bf16": (c_int, [c_vp, c_i64, c_vp, c_vp]),
    "insmos_register_split_weights": (c_int, [c_vp, c_vp]),
    "insmos_dense_nbr2d": (c_int, [c_int, c_int, c_vp, c_vp]),
    "insmos_sparse_to_bev": (c_int, [c_vp, c_int, c_int, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    "insmos_center_decode_select_ws_bytes": (c_sz, [c_i64]),
    "insmos_center_decode_select": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_f32, c_f32, c_f32, c_f32,
                                            c_f32, c_f32, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_nms_ws_bytes": (c_sz, [c_int]),
    "insmos_nms_rotated_bev": (c_int, [c_vp, c_vp, c_int, c_f32, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "insmos_iou_bev": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "insmos_iou3d": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "insmos_gather_preds": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "insmos_boxes_to_onehot_scratch_ints": (c_sz, [c_int, c_i64]),
    "insmos_copy_cols": (c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_int, c_vp]),
    "insmos_boxes_to_onehot": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_f32, c_f32, c_vp, c_i64, c_int, c_int,
                                       c_int, c_vp, c_int, c_vp, c_vp]),
    "insmos_gather_rows": (c_int, [c_vp, c_int, c_int, c_vp, c_i64, c_vp, c_int, c_vp]),
    "insmos_build_current_points": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_i64, c_vp, c_int, c_vp]),
    "insmos_fill_cols": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_f32, c_vp]),
    "insmos_stack_scan": (c_int, [c_vp, c_i64, c_vp, c_f32, c_vp, c_int, c_vp]),
    "insmos_output_stage": (c_int, [c_vp, c_int, c_i64, c_int, c_u32, c_vp, c_vp, c_vp, c_vp]),
    "insmos_confusion3": (c_int, [c_vp, c_int, c_vp, c_i64, c_int, c_u32, c_vp, c_vp]),
    "insmos_points_in_instance_boxes": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_f32, c_int, c_int, c_vp, c_vp, c_vp]),
    "insmos_instance_stats": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_vp]),
    "insmos_instance_relabel": (c_int, [c_vp, c_int, c_int, c_vp, c_i64, c_int, c_vp, c_vp]),
    "insmos_pack_weights_device": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "insmos_sparse_conv_backward_weight_ws_floats": (c_sz, [c_i64, c_int, c_int, c_int]),
    "insmos_sparse_conv_backward_weight": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_int, c_int, c_vp, c_int, c_i64, c_vp, c_int,
                                                   c_vp, c_vp]),
    "insmos_col_sum_ws_floats": (c_sz, [c_i64, c_int]),
    "insmos_col_sum": (c_int, [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp, c_vp]),
    "insmos_batchnorm_seg_ws_floats": (c_sz, [c_int, c_int, c_int]),
    "insmos_batchnorm_seg_forward": (c_int, [c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_f32, c_int, c_vp,
                                             c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp]),
    "insmos_batchnorm_seg_backward": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp,
                                              c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_batchnorm_seg_recompute_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "insmos_batchnorm_seg_backward_x": (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp,
                                                c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_batchnorm_ws_floats": (c_sz, [c_i64, c_int]),
    "insmos_batchnorm_train_forward": (c_int, [c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_f32, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "insmos_batchnorm_train_backward": (c_int, [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_i64, c_vp, c_vp, c_int, c_vp, c_int, c_vp,
                                                c_vp, c_vp, c_vp, c_vp]),
    "insmos_mos_loss_ws_floats": (c_sz, [c_i64]),
    "insmos_mos_loss": (c_int, [c_vp, c_int, c_vp, c_i64, c_int, c_u32, c_vp, c_vp, c_vp, c_int, c_vp, c_vp]),
    "insmos_center_assign_targets": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_f64, c_f64, c_int, c_f32, c_f32, c_int,
                                             c_f64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "insmos_center_head_loss_ws_floats": (c_sz, [c_i64, c_int]),
    "insmos_center_head_loss": (c_int, [c_vp, c_int, c_vp, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_f32,
                                        c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "insmos_ctx_create": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "insmos_ctx_destroy": (c_int, [c_vp]),
    "insmos_forward_window": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_sz, c_vp, c_vp]),
}


class ConvW(ctypes.Structure):  # InsmosConvW
    _fields_ = [("w", c_vp), ("b", c_vp), ("K", ctypes.c_int32), ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class NetCfg(ctypes.Structure):  # InsmosNetCfg
    _fields_ = [("w0_const", c_vp), ("b0_const", c_vp), ("nbr_bev", c_vp),
                ("vs", c_f32 * 3), ("dt", c_f32), ("range", c_f32 * 6), ("score_thresh", c_f32), ("nms_thresh", c_f32),
                ("out_factor", c_f32), ("tvs", c_f32 * 2),
                ("in_ch", ctypes.c_int32), ("ncls", ctypes.c_int32), ("max_voxels", ctypes.c_int32),
                ("max_points", ctypes.c_int32), ("shape", (ctypes.c_int32 * 3) * 6),
                ("bevD", ctypes.c_int32), ("bevH", ctypes.c_int32), ("bevW", ctypes.c_int32), ("nbev", ctypes.c_int32),
                ("n_bev_layers", ctypes.c_int32), ("up_ch", ctypes.c_int32), ("head_ld", ctypes.c_int32),
                ("pre_max", ctypes.c_int32), ("post_max", ctypes.c_int32), ("quirk_exact", ctypes.c_int32)]


class RankJob(ctypes.Structure):  # InsmosRankJob
    _fields_ = [("out_coords", c_vp), ("bits", c_vp), ("blk_incl", c_vp), ("in_perm", c_vp), ("nbr", c_vp), ("mask16", c_vp),
                ("in_shape", c_vp), ("delta", c_vp), ("mul", c_vp), ("div", c_vp), ("n_out", c_i64), ("K", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class ForwardOut(ctypes.Structure):  # InsmosForwardOut
    _fields_ = [("me_voxels", c_i64 * 4), ("n_cur", c_i64), ("unet_voxels", c_i64 * 5), ("n_candidates", c_i64),
                ("n_boxes", c_i64), ("n_out_of_window", c_i64), ("logits_off", c_i64), ("boxes_off", c_i64),
                ("scores_off", c_i64), ("labels_off", c_i64), ("arena_needed", c_i64), ("cur_points_off", c_i64),
                ("batch", c_i64)]

_lib = None


class InsmosHipError(RuntimeError):
    pass


def load():
    """dlopen libinsmos_hip.so (built by __graft_entry__.build()).  Raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise InsmosHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  insmos_amd has no CPU fallback.")
        # torch first: its bundled HIP runtime must be the one in the process' global symbol scope before this library
        # is mapped, so that both talk to the SAME runtime (streams and device pointers cross the boundary).  Loading
        # libinsmos_hip.so before torch binds it to /opt/rocm's runtime instead and every launch fails (hipErrorNoDevice).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_ERR = {-1: "INSMOS_EINVAL (bad argument)", -2: "INSMOS_EHIP (HIP runtime error)", -3: "INSMOS_EWORKSPACE",
        -4: "INSMOS_EBATCH (launch set too large)"}


def check(rc, what):
    if rc != 0:
        extra = ""
        if rc == -2:
            extra = f", hipError={load().insmos_last_hip_error()}"
        raise InsmosHipError(f"{what} failed: {_ERR.get(rc, rc)}{extra}")
