"""ctypes binding of libfsgs.so (C-ABI declared in include/fsgs.h).

The product path has NO fallback: if the shared library is missing or a call fails, this
module raises.  Pointers are raw device addresses taken from torch tensors; torch is used for
memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (FSGS_LIB: A/B runs against another build of the same C-ABI, e.g. `make OUT=../libfsgs_x.so BUILD=build_x EXTRA=-D...`)
LIB_PATH = os.environ.get("FSGS_LIB") or os.path.join(_HERE, "libfsgs.so")

_i, _i64, _f, _p, _sz = C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/fsgs.h one to one
class AdamGroups(C.Structure):
    """fsgs_adam_groups of include/fsgs.h: one Adam step over up to 8 tensors as an argument block."""
    _fields_ = [("n_groups", C.c_int), ("params", C.c_void_p * 8), ("grads", C.c_void_p * 8),
                ("exp_avg", C.c_void_p * 8), ("exp_avg_sq", C.c_void_p * 8),
                ("half_mirror", C.c_void_p * 8), ("numel", C.c_int64 * 8), ("lr", C.c_float * 8),
                ("step", C.c_int), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_float)]


class RefineRules(C.Structure):
    """fsgs_refine_rules of include/fsgs.h: the schedule scalars and thresholds of one refinement / cull."""
    _fields_ = [("densify", C.c_int32), ("n_samples", C.c_int32), ("max_hw", C.c_float),
                ("densify_grad_thresh", C.c_float), ("densify_size_thresh", C.c_float), ("check_screen", C.c_int32),
                ("split_screen_size", C.c_float), ("cull_alpha_thresh", C.c_float), ("check_big", C.c_int32),
                ("cull_scale_thresh", C.c_float), ("cull_screen_size", C.c_float), ("hull_center", C.c_float * 3),
                ("hull_close", C.c_float), ("hull_lo", C.c_float), ("hull_hi", C.c_float), ("n_boxes", C.c_int32),
                ("grid_origin", C.c_float * 3), ("grid_inv_cell", C.c_float), ("grid_dims", C.c_int32 * 3)]


class StepPlan(C.Structure):
    """fsgs_step_plan of include/fsgs.h, field for field (tests/test_host_logic.py compares every offset with the C
    compiler's): one training step's launches as one argument block (fusionsense_amd/express.py fills it)."""
    _fields_ = [
        ("n", _i), ("sh_degree", _i), ("width", _i), ("height", _i), ("tile_width", _i), ("tile_height", _i),
        ("capacity", _i64),
        ("means", _p), ("quats", _p), ("log_scales", _p), ("opac_logit", _p), ("features_dc", _p), ("features_rest", _p),
        ("viewmat", _p), ("K", _p), ("campos", _p), ("c2w", _p), ("background", _p),
        ("binarise", _i), ("binary_threshold", _f),
        ("scales_exp", _p), ("opac_sig", _p), ("radii", _p), ("means2d", _p), ("depths", _p), ("conics", _p),
        ("tiles_per_gauss", _p), ("isect_offsets", _p),
        ("bucket_base", _p), ("tile_cursor", _p),
        ("buckets", _p), ("bucket_words", _i64), ("next_bucket_base", _p),
        ("growth", _f), ("slack", _i), ("mapped", _p),
        ("packed", _p), ("normals_world", _p), ("zero_cells", _p), ("n_zero", _i),
        ("tile_order", _p),
        ("payload", _p), ("long_flag", _p), ("rel_gate", _i),
        ("render", _p), ("alphas", _p), ("last_ids", _p), ("render_extra", _p),
        ("records", _p), ("n_rec", _p), ("seg_state", _p), ("seg_split", _p),
        ("max_last", _p),
        ("tail_scratch", _p), ("tail_scratch_bytes", _i64), ("tail_items", _i), ("handoff_records", _i),
        ("handoff_rel_len", _i), ("tail_epoch", _i64),
        ("bwd_queue", _p), ("bwd_queue_items", _i),
        ("rgb", _p), ("depth", _p), ("normal", _p), ("n_cells", _i),
        ("gt_rgb", _p), ("gt_depth", _p), ("gt_normal", _p), ("seed", _p),
        ("g_depth", _f), ("g_normal", _f), ("aux_partial", _p), ("v_depth_img", _p), ("v_normal_img", _p),
        ("order_counters", _p), ("bwd_order", _p), ("order_shift", _i),
        ("ssim_maps", _p), ("ssim_sums", _p), ("ssim_rows", _i64), ("aux_rows", _i64),
        ("g_l1", _f), ("g_ssim", _f), ("ssim_lambda", _f), ("v_rgb", _p), ("loss_out", _p),
        ("loss_kind", _i), ("mask", _p), ("sensor_depth", _p), ("depth_tol", _f), ("w_aux", _f * 7), ("fa_flags", _i),
        ("fa_partial", _p), ("fa_rows", _i64), ("ms_partial", _p), ("ms_rows", _i64), ("g_min", _f),
        ("n_touch", _i), ("touch_idx", _p), ("touch_normals", _p), ("touch_partial", _p), ("touch_rows", _i64),
        ("g_touch", _f),
        ("v_packed", _p), ("replica_rows", _i64), ("dispatch_stride", _i),
        ("absgrad", _p), ("xys_grad_norm", _p), ("vis_counts", _p), ("max_2Dsize", _p),
        ("inv_max_hw", _f), ("frozen", _p),
        ("adam", AdamGroups), ("min_scale_g", _f), ("gsb_flags", _i),
        ("g_means", _p), ("g_log_scales", _p), ("g_quats", _p), ("g_features_dc", _p), ("g_features_rest", _p),
        ("g_opac_logit", _p),
        ("ev_before", _p * 9), ("ev_after", _p * 9),
        ("armed", _i), ("wait_ns", _i64),
    ]


# == FSGS_ABI_VERSION of include/fsgs.h as of the SIGNATURES table below: load() refuses any other library (a stale
# build — the .so files are git-ignored and travel separately, A/B builds come in through FSGS_LIB — would read a stream
# pointer as a flag or write past a buffer that has since grown)
ABI_VERSION = 10

SIGNATURES = {
    "fsgs_version": (_i, []),
    "fsgs_abi_version": (_i, []),
    "fsgs_grad_replica_lines": (_i, []),
    "fsgs_error_string": (C.c_char_p, [_i]),
    "fsgs_last_hip_error": (_i, []),
    "fsgs_project_fwd": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p]),
    "fsgs_project_bwd": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_sh_fwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_sh_bwd": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "fsgs_scan_scratch_bytes": (_sz, [_i64]),
    "fsgs_isect_count": (_i, [_i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _sz, C.POINTER(_i64), C.POINTER(_i64), _p]),
    "fsgs_isect_emit": (_i, [_i, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "fsgs_sort_scratch_bytes": (_sz, [_i64]),
    "fsgs_sort_pairs": (_i, [_i64, _p, _p, _p, _p, _i, _p, _sz, C.POINTER(_i), _p]),
    "fsgs_isect_offset_encode": (_i, [_i64, _p, _i, _i, _i, _p, _p]),
    "fsgs_raster_fwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i64, _p, _p, _p, _p]),
    "fsgs_raster_bwd": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_isect_count_live": (_i, [_i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _sz, C.POINTER(_i64), _p]),
    "fsgs_isect_emit_live": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "fsgs_tile_sort_max_tiles": (_i, []),
    "fsgs_tile_sort_scratch_bytes": (_sz, [_i64, _i]),
    "fsgs_tile_sort": (_i, [_i64, _p, _p, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "fsgs_bin_live_max_tiles": (_i, []),
    "fsgs_bin_live_table_bytes": (_sz, [_i, _i, _i, _i]),
    "fsgs_bin_live_count": (_i, [_i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _sz, _p, _p]),
    "fsgs_project_bin_live_count_sh_pack": (_i, [_i, _p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _sz, _p,
                                                 _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "fsgs_project_bin_live_count_sh_pack_h16": (_i, [_i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _sz, _p,
                                                     _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "fsgs_project_bin_live_count": (_i, [_i, _p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _sz, _p, _p]),
    "fsgs_bin_live_emit": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i64, _p, _p, _p, _p, _i, _p]),
    "fsgs_project_bin_live_fill_sh_pack": (_i, [_i, _p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p,
                                                _p, _p, _p, _i64, _p, _f, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p]),
    "fsgs_bin_live_sort_buckets": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _p, _p]),
    "fsgs_bin_live_split_scratch_bytes": (_sz, [_i, _i, _i, _i64]),
    "fsgs_bin_live_emit_split": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i64, _p, _p, _p, _sz, _p, _p, _p]),
    "fsgs_set_lazy_sh_min_n": (_i, [_i]),
    "fsgs_tile_zcut_recheck": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "fsgs_tile_zcut_update": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _p]),
    "fsgs_quad_stream_capacity": (_i64, [_i, _i, _i, _i64]),
    "fsgs_quad_seg_slots": (_i64, [_i, _i, _i, _i64]),
    "fsgs_live_pack": (_i, [_i, _i64, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "fsgs_raster_quad_max_cells": (_i, []),
    "fsgs_live_pack_normals": (_i, [_i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "fsgs_live_payload": (_i, [_p, _p, _i64, _p, _i64, _i, _i, _p, _p]),
    "fsgs_raster_fwd_quad": (_i, [_i, _i, _p, _p, _p, _i64, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i64, _i, _i, _i, _p, _i64, _p, _p, _i, _p, _p]),
    "fsgs_raster_fwd_tail_scratch_bytes": (_i64, [_i]),
    "fsgs_raster_fwd_tail_error": (_i, [_p, _p]),
    "fsgs_set_bwd_dispatch_stride": (_i, [_i]),
    "fsgs_raster_bwd_quad": (_i, [_i, _i, _p, _p, _p, _i64, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _p, _p, _i, _p, _p]),
    "fsgs_raster_bwd_quad_images": (_i, [_p, _p, _p, _i64, _i, _i, _i, _i] + [_p] * 10 + [_i, _p, _i64, _i, _p, _p, _i, _p, _p]),
    "fsgs_campos_from_viewmats": (_i, [_i, _p, _p, _p]),
    "fsgs_raster_unpack_grads": (_i, [_i64, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_sh_fwd_split": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_sh_bwd_split": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i64, _p]),
    "fsgs_sh_fwd_pack": (_i, [_i, _i, _i] + [_p] * 15 + [_i, _p]),
    "fsgs_sh_bwd_colors": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i64, _p]),
    "fsgs_sh_coeff_grad": (_i, [_i, _i, _i, _i, _p, _p, _f, _p, _p, _p]),
    "fsgs_sh_coeff_grad_adam": (_i, [_i, _i, _i, _i, _p, _p, _f, _p, _p, _p, _f, _p, _p, _p, _f, _i, C.c_double, C.c_double, _f, _p]),
    "fsgs_project_fwd_act": (_i, [_i, _i, _p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_gauss_sh_bwd_adam": (_i, [_i, _i] + [_p] * 11 + [_i, _i, _f] + [_p] * 7 + [_f, _p, _i64, _p, _f, _i, _p]),
    "fsgs_gauss_sh_bwd_adam_h16": (_i, [_i, _i] + [_p] * 11 + [_i, _i, _f] + [_p] * 7 + [_f, _p, _i64, _p, _f, _i, _p]),
    "fsgs_gauss_sh_bwd_h16": (_i, [_i, _i] + [_p] * 11 + [_i, _i, _f] + [_p] * 13 + [_f, _p, _i64, _i, _p]),
    "fsgs_gauss_sh_bwd": (_i, [_i, _i] + [_p] * 11 + [_i, _i, _f] + [_p] * 14 + [_f, _p, _i64, _i, _p]),
    "fsgs_gaussian_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _p]),
    "fsgs_activate_fwd": (_i, [_i, _p, _p, _p, _p, _p]),
    "fsgs_activate_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_epilogue_fwd": (_i, [_i64, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "fsgs_epilogue_loss_fwd": (_i, [_i64, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p]),
    "fsgs_epilogue_fwd_order": (_i, [_i64, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p,
                                     _p, _p, _p, _p, _i, _i, _i, _p]),
    "fsgs_epilogue_bwd": (_i, [_i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_normals_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_normals_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_densify_stats": (_i, [_i, _p, _p, _f, _p, _p, _p, _p]),
    "fsgs_mask_scan": (_i, [_i64, _p, _p, _p, _sz, _p]),
    "fsgs_compact_rows": (_i, [_i64, _i, _p, _p, _p, _p, _p]),
    "fsgs_compact_rows_multi": (_i, [_i, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_points_in_boxes": (_i, [_i64, _p, _i, _p, _p, _p]),
    "fsgs_nearest_point": (_i, [_i, _p, _i, _p, _p, _p, _p, _i, _p]),
    "fsgs_nearest_point_words": (_i, [_i, _p, _i, _p, _p, _f, _p, _p]),
    "fsgs_knn_points": (_i, [_i64, _p, _i, _p, _i, _i, _p, _p]),
    "fsgs_refine_book_ints": (_i64, [_i64]),
    "fsgs_refine_mark": (_i, [_i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_refine_move": (_i, [_i64, _i, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i, _p, _p, _p, _p, _p,
                              _i, _p]),
    "fsgs_split_samples": (_i, [_i64, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_ssim_l1_num_partials": (_i64, [_i, _i]),
    "fsgs_ssim_l1_fwd": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_adam_step": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, C.c_double, C.c_double, _f, _p]),
    "fsgs_aux_l1_fwd": (_i, [_i64, _p, _p, _p, _p, _p, _p]),
    "fsgs_aux_l1_bwd": (_i, [_i64, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p]),
    "fsgs_aux_l1_fwd_bwd": (_i, [_i64, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p]),
    "fsgs_ssim_l1_bwd": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p]),
    "fsgs_ssim_l1_bwd_combine": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _f, _f, _p, _i, _p, _p, _p, _f, _p, _p]),
    "fsgs_loss_combine": (_i, [_i, _p, _p, _p, _f, _p, _p]),
    "fsgs_project_bin_live_count_h16": (_i, [_i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _sz, _p, _p]),
    "fsgs_project_fwd_act_h16": (_i, [_i, _i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_sh_fwd_pack_h16": (_i, [_i, _i, _i] + [_p] * 15 + [_i, _p]),
    "fsgs_sh_bwd_split_h16": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i64, _p]),
    "fsgs_gaussian_bwd_h16": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _i64, _p]),
    "fsgs_adam_step_h16": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _i, C.c_double, C.c_double, _f, _p]),
    "fsgs_loss_combine_cols": (_i, [_i, _p, _p, _p, _p, _f, _p, _p]),
    "fsgs_ssim_l1_fwd_masked": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "fsgs_ssim_l1_bwd_masked": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _i, _p, _p, _p, _p, _f, _p, _p]),
    "fsgs_fusion_aux_num_partials": (_i64, [_i, _i]),
    "fsgs_fusion_aux_loss": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _p]),
    "fsgs_fusion_aux_loss_riders": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i,
                                         _i, _p, _p, _i, _p, _p, _p, _p, _p]),
    "fsgs_normals_from_depth": (_i, [_i, _i, _p, _p, _f, _f, _f, _f, _p, _p]),
    "fsgs_normal_cosine_num_partials": (_i64, [_i]),
    "fsgs_normal_cosine_loss": (_i, [_i, _i, _p, _p, _p, _i, _f, _p, _p, _p, _p]),
    "fsgs_depth_valid_counts": (_i, [_i, _i, _p, _p, _f, _p, _p]),
    "fsgs_min_scale_loss": (_i, [_i, _p, _f, _p, _p, _p, _p]),
    "fsgs_touch_normal_sqerr": (_i, [_i, _p, _p, _p, _p, _p]),
    "fsgs_step_forward": (_i, [C.POINTER(StepPlan), _p]),
    "fsgs_step_backward": (_i, [C.POINTER(StepPlan), _i64, C.POINTER(_i64), _p]),
    "fsgs_step_run": (_i, [C.POINTER(StepPlan), _i64, C.POINTER(_i64), _p]),
    "fsgs_step_plan_bytes": (_i64, []),
}

_lib: Optional[C.CDLL] = None


class FsgsError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libfsgs.so (built in-tree by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FsgsError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C fusionsense_amd/csrc`). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    try:
        lib.fsgs_abi_version.restype = C.c_int
        found = int(lib.fsgs_abi_version())
    except AttributeError:
        found = None
    if found != ABI_VERSION:
        raise FsgsError(f"{LIB_PATH} has C-ABI version {found}, this binding was written against {ABI_VERSION} "
                        "(include/fsgs.h: FSGS_ABI_VERSION): rebuild it (`make -C fusionsense_amd/csrc`)")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t: Optional[torch.Tensor]):
    """Raw device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "fsgs: tensor must be contiguous"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device):
    """The hipStream_t of torch's CURRENT stream on ``device`` (what every launch is enqueued on).  Through torch's raw
    accessor where it exists: building a ``torch.cuda.Stream`` object per launch cost ~6 us of host time — 80-90 us per
    step over the 13-15 launches of a frame (cProfile of the shim route, round 4)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.fsgs_error_string(rc).decode()
        raise FsgsError(f"{what} failed: {msg} (code {rc}, hipError {lib.fsgs_last_hip_error()})")


def require_gpu_tensor(t: torch.Tensor, name: str, dtype=torch.float32) -> None:
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (there is no CPU path); got device {t.device}")
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
