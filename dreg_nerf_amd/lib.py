"""ctypes binding of the C-ABI library (include/dreg_nerf.h).  No CPU fallback: if the shared object is
missing or a symbol fails, the product path raises."""
import ctypes
import os
from ctypes import c_float, c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdreg_nerf_hip.so")

DT_BF16 = 0
DT_F32 = 1

_lib = None


class DregError(RuntimeError):
    pass


def _sig(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


P, I, F, Z = c_void_p, c_int, c_float, c_size_t

SIGNATURES = {
    # conv.hip
    "dreg_conv3d_igemm": (I, [P, P, P, P, P] + [I] * 18 + [I, I, P]),
    "dreg_conv3d_igemm_workspace_bytes": (Z, [I] * 15),
    "dreg_conv3d_igemm_ws": (I, [P, P, P, P, P] + [I] * 18 + [I, I, P, Z, P]),
    "dreg_conv3d_igemm_occ": (I, [P, P, P, P, P] + [I] * 18 + [I, I, P, Z, P, P]),
    "dreg_conv3d_igemm_bnstats": (I, [P, P, P, P, P] + [I] * 17 + [P, Z, P, P, P]),
    "dreg_conv3d_kpad": (I, [I, I, I]),
    "dreg_gemm_f32_desc_bytes": (I, []),
    "dreg_gemm_f32_batched": (I, [P, I, I, P, P]),
    "dreg_infonce_wsym": (I, [P, P, I, P]),
    "dreg_conv3d_igemm_defer": (I, [P, P, P, P, P] + [I] * 18 + [P, Z, P, P, P, P]),
    "dreg_bn3d_fwd_ex": (I, [P] * 10 + [I, I, I, F, F, I, I, I, P, P, I, P, P]),
    "dreg_bn3d_bwd_ex": (I, [P] * 11 + [I] * 6 + [P, P, P, P]),
    "dreg_exec_default_opts": (None, [P]),
    "dreg_exec_guard_bands": (I, [P]),
    "dreg_exec_guard_check": (I, [P, P, P, P]),
    "dreg_exec_guard_describe": (I, [P, I, P, I]),
    "dreg_exec_guard_last": (I, [P, P, P, P]),
    "dreg_guard_fill": (I, [P, P, I, I, P]),
    "dreg_guard_scan": (I, [P, P, I, I, P, P]),
    "dreg_exec_create_opts": (P, [P, I, P, I, P, I, P]),
    "dreg_ps_set_group_wgrad": (None, [P, I]),
    "dreg_conv_get_glds": (I, []),
    "dreg_conv3d_dgrad_s2": (I, [P, P, P] + [I] * 11 + [P]),
    "dreg_conv3d_dgrad_s2_acc": (I, [P, P, P] + [I] * 11 + [P]),
    "dreg_pack_conv_weight": (I, [P, P, I, I, I, I, I, I, P]),
    "dreg_pack_conv_weights_batched": (I, [P, I, I, I, P, P]),
    "dreg_conv3d_wgrad_splits": (I, [I] * 8),
    "dreg_conv3d_wgrad_workspace_bytes": (Z, [I] * 8),
    "dreg_conv3d_wgrad": (I, [P, P, P, P, Z] + [I] * 16 + [P]),
    "dreg_conv3d_wgrad_occ": (I, [P, P, P, P, Z] + [I] * 16 + [P, P]),
    "dreg_conv3d_igemm_rows": (I, [P] * 5 + [P, I] + [I] * 18 + [I, P]),
    "dreg_conv3d_wgrad_rows": (I, [P, P, P, P, Z] + [P, I] + [I] * 14 + [P]),
    "dreg_conv3d_wgrad_partials": (I, [P, P, P, Z, P, I] + [I] * 13 + [P, P]),
    "dreg_wgrad_reduce_blocks": (I, [I, I, I, I]),
    "dreg_wgrad_reduce_batched": (I, [P, I, I, I, P]),
    # conv_halo.hip
    "dreg_conv3_halo_supported": (I, [I] * 6),
    "dreg_conv3_halo_use": (I, [I] * 9),
    "dreg_conv3_halo_pack_bytes": (Z, [I]),
    "dreg_pack_conv_weight_halo": (I, [P, P, I, I, I, P]),
    "dreg_conv3_halo": (I, [P, P, P, P, P] + [I] * 10 + [P]),
    "dreg_conv3_halo_pack_bytes_n": (Z, [I, I]),
    "dreg_conv3_halo_n": (I, [P, P, P, P, P] + [I] * 11 + [P]),
    "dreg_conv3_halo_n_bnstats": (I, [P, P, P, P, P] + [I] * 10 + [P, P, P]),
    "dreg_conv3d_wgrad_variant": (I, [I] * 10),
    "dreg_conv3d_igemm_variant": (I, [I] * 17),
    "dreg_conv3d_wgrad_group_fill": (I, [P, P, P, P, Z] + [I] * 12 + [P, P]),
    "dreg_bn_small_in_regs": (I, [I, I, I, I]),
    "dreg_wgrad_group_desc_bytes": (I, []),
    "dreg_linear_wgrad_group_fill": (I, [P, P, P, P, Z, I, I, I, P, P]),
    "dreg_wgrad_group_launch": (I, [P, I, I, I, P]),
    # fpn_ops.hip
    "dreg_bn_num_chunks": (I, [I]),
    "dreg_bn_relu_maxpool_fwd": (I, [P] * 10 + [I] * 8 + [F, F, I, I, P]),
    "dreg_bn_relu_maxpool_bwd": (I, [P] * 10 + [I] * 10 + [P]),
    "dreg_sparse_stem_workspace_floats": (Z, [I] * 5),
    "dreg_sparse_stem_fwd": (I, [P, P, I, P, I] + [P] * 12 + [I] * 8 + [F, F, I, I, P]),
    "dreg_sparse_stem_bwd": (I, [P] * 6 + [I, P, I] + [P] * 7 + [I] * 10 + [P]),
    "dreg_bn3d_fwd": (I, [P] * 10 + [I, I, I, F, F, I, I, I, P]),
    "dreg_bn3d_bwd": (I, [P] * 11 + [I, I, I, I, I, I, P]),
    "dreg_bn_small": (I, [I, I, I, I]),
    "dreg_bn3d_fwd_defer_update": (I, [P] * 10 + [I, I, I, F, F, I, I, I, P, P, P]),
    "dreg_bn3d_fwd_from_sums": (I, [P] * 10 + [I, I, I, I, F, F, I, I, P]),
    "dreg_bn3d_bwd_defer_params": (I, [P] * 11 + [I, I, I, I, I, I, P, P, P]),
    "dreg_bn_running_update_batched": (I, [P, I, I, I, F, P]),
    "dreg_bn_param_grad_batched": (I, [P, I, I, I, I, P]),
    "dreg_maxpool3d_fwd": (I, [P, P, P] + [I] * 9 + [P]),
    "dreg_maxpool3d_bwd": (I, [P, P, P] + [I] * 9 + [P]),
    "dreg_maxpool3d_bwd_acc": (I, [P, P, P] + [I] * 10 + [P]),
    "dreg_downsample_sum": (I, [P, P] + [I] * 9 + [P]),
    "dreg_downsample_sum_rows": (I, [P, P, P] + [I] * 9 + [P]),
    "dreg_colsum_workspace_bytes": (Z, [Z, I]),
    "dreg_colsum": (I, [P, P, P, Z, I, I, I, P]),
    "dreg_trilinear_gather_fwd": (I, [P, P, P, P] + [I] * 10 + [P]),
    "dreg_trilinear_gather_bwd": (I, [P, P, P, P] + [I] * 8 + [P]),
    "dreg_cast_from_f32": (I, [P, P, Z, I, P]),
    "dreg_fill_zero": (I, [P, Z, P]),
    "dreg_active_sets_workspace_bytes": (Z, [I] * 4),
    "dreg_active_sets": (I, [P, P] + [I] * 8 + [P, P, P, P, Z, P]),
    "dreg_active_sets_level2_workspace_bytes": (Z, [I] * 4),
    "dreg_active_sets_level2": (I, [P] + [I] * 7 + [P, P, P, Z, P]),
    "dreg_conv_rows_workspace_bytes": (Z, [I, I, I, I]),
    "dreg_conv_rows": (I, [P, P, I, I, I, I, I, I, I, I, I, I, I, P, P, P, Z, P]),
    "dreg_trilinear_gather_bwd_rows": (I, [P, P, P, P, I, P, P, P] + [I] * 10 + [P]),
    "dreg_colsum_rows": (I, [P, P, I, P, P, I, I, I, P]),
    "dreg_trilinear_gather_bwd_gather": (I, [P, P, P, P, I, P, P] + [I] * 10 + [P]),
    "dreg_trilinear_gather_bwd_gather_rows_only": (I, [P, P, P, P, I, P, P] + [I] * 10 + [P]),
    "dreg_trilinear_gather_bwd_gather_seg": (I, [P, P, P, P, P, I, P, P] + [I] * 11 + [P]),
    "dreg_gather_segment_mean": (I, [P, P, P, I, P, P, P, P] + [I] * 9 + [P]),
    "dreg_zero_rows": (I, [P, P, I, I, I, P]),
    "dreg_add_inplace": (I, [P, P, Z, I, P]),
    "dreg_pack_rgba_grids": (I, [P, P, I, I, I, I, I, P]),
    "dreg_pack_rgba_grids_occ": (I, [P, P, P, I, I, I, I, I, P]),
    "dreg_pack_rgba_sparse": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "dreg_pack_rgba_sparse_occ": (I, [P, P, P, P, P, I, I, I, I, I, I, P]),
    "dreg_conv_row_occupancy": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "dreg_gather_grid_xyz": (I, [P, P, P, P, I, I, I, I, P]),
    # executor.hip
    "dreg_exec_create": (P, [P, I, P, I, P, I]),
    "dreg_exec_destroy": (None, [P]),
    "dreg_exec_op_halo": (I, [P, I]),
    "dreg_exec_arena_bytes": (Z, [P]),
    "dreg_exec_pack_bytes": (Z, [P]),
    "dreg_exec_num_packs": (I, [P]),
    "dreg_exec_tensor_offset": (Z, [P, I]),
    "dreg_exec_output_slot": (I, [P]),
    "dreg_exec_pack_rows": (I, [P]),
    "dreg_exec_export_pack_table": (I, [P, P, P, P]),
    "dreg_exec_repack": (I, [P, P, P, P]),
    "dreg_exec_set_overlap": (None, [P, I]),
    "dreg_exec_set_input_row_occupancy": (None, [P, P]),
    "dreg_exec_set_timing": (None, [P, I]),
    "dreg_exec_read_timings": (I, [P, P, P, I]),
    "dreg_exec_forward": (I, [P, P, Z, P, P, P, I, I, P]),
    "dreg_exec_backward": (I, [P, P, Z, P, P, P, P, I, P, P]),
    "dreg_exec_backward_range": (I, [P, P, Z, P, P, P, P, I, P, P, I, I, I]),
    # attention.hip
    "dreg_mha_fwd": (I, [P] * 5 + [I] * 7 + [F, I, P]),
    "dreg_mha_bwd": (I, [P] * 10 + [I] * 7 + [F, I, P]),
    "dreg_corr_attention_fwd": (I, [P] * 5 + [I] * 3 + [F, I, P]),
    "dreg_corr_attention_bwd": (I, [P] * 9 + [I] * 3 + [F, I, P]),
    "dreg_mha_varlen_fwd": (I, [P] * 6 + [I] * 9 + [F, I, P]),
    "dreg_mha_varlen_bwd": (I, [P] * 11 + [I] * 9 + [F, I, P]),
    "dreg_corr_attention_varlen_fwd": (I, [P] * 6 + [I] * 5 + [F, I, P]),
    "dreg_corr_attention_varlen_bwd": (I, [P] * 10 + [I] * 5 + [F, I, P]),
    # pointset.hip
    "dreg_layernorm_fwd": (I, [P] * 6 + [I, I, F, I, P]),
    "dreg_layernorm_bwd_workspace_bytes": (Z, [I]),
    "dreg_layernorm_bwd": (I, [P] * 8 + [I] * 5 + [P]),
    "dreg_layernorm_bwd_add": (I, [P] * 9 + [I] * 4 + [P]),
    "dreg_layernorm_fwd2": (I, [P, P, P, P, I, P, P, P, I, I, F, P]),
    "dreg_layernorm_bwd_parts": (I, [P] * 10 + [I, I, I, P]),
    "dreg_layernorm_bwd_blocks": (I, [I]),
    "dreg_layernorm_bwd_final_batched": (I, [P, I, P]),
    "dreg_overlap_bwd_acc": (I, [P, P, P, P, P, P, P, P, I, P, I, P]),
    "dreg_colsum_rows_per_chunk": (I, [Z]),
    "dreg_colsum_batched": (I, [P, I, I, I, P]),
    # conv_brick.hip
    "dreg_brick_supported": (I, [I] * 6),
    "dreg_conv3_brick_pack_bytes": (Z, [I, I]),
    "dreg_pack_conv_weight_brick": (I, [P, P, I, I, I, P]),
    "dreg_brick_tiles_workspace_bytes": (Z, [I] * 4),
    "dreg_brick_tiles_build": (I, [P, I, I, I, I, I, I, P, Z, P, P, P, P, P, P]),
    "dreg_conv3_brick": (I, [P, P, P, P, P, P, I, P, P, P] + [I] * 11 + [P]),
    # pointset_exec.hip
    "dreg_ps_num_params": (I, []),
    "dreg_ps_num_linears": (I, []),
    "dreg_ps_create": (P, [P]),
    "dreg_ps_destroy": (None, [P]),
    "dreg_ps_set_fuse": (None, [P, I]),
    "dreg_ps_set_timing": (None, [P, I]),
    "dreg_ps_read_timings": (I, [P, P, P, I]),
    "dreg_ps_arena_bytes": (Z, [P, I]),
    "dreg_ps_forward": (I, [P, P, Z, P, P, P, P, P, P, I, I, I, P, P, P, P]),
    "dreg_ps_backward": (I, [P, P, Z, P, P, P, P, P, P, I, I, I, P, P, P, P, P, P, P, P, P, I]),
    "dreg_posenc_sine": (I, [P, P, I, F, F, P]),
    "dreg_overlap_fwd": (I, [P, P, P, P, I, P]),
    "dreg_overlap_bwd_workspace_bytes": (Z, [I]),
    "dreg_overlap_bwd": (I, [P] * 8 + [I, P]),
    "dreg_relu_bwd": (I, [P, P, P, Z, I, I, I, P]),
    "dreg_weighted_kabsch": (I, [P, P, P, P, I, I, F, P]),
    "dreg_weighted_kabsch_pairs": (I, [P, P, P, P, P, I, I, I, F, P]),
    # losses.hip
    "dreg_reg_point_losses": (I, [P] * 10 + [I, I, I, I, F, F, F, P]),
    "dreg_infonce_nn": (I, [P] * 7 + [I, I, F, P]),
    "dreg_infonce_rows": (I, [P] * 10 + [I, I, F, F, I, P]),
    "dreg_reg_losses_final": (I, [P] * 6 + [I, I, F, F, F, F, F, P]),
    "dreg_nerf_cont_deferred": (I, [P] * 5 + [I, I, I, F, F, F, F, P]),
    "dreg_halfspace_labels": (I, [P, P, P, P, I, I, P]),
    "dreg_voxel_downsample_workspace_bytes": (Z, [I]),
    "dreg_voxel_downsample_fwd": (I, [P] * 11 + [Z, I, I, I, F, P]),
    "dreg_voxel_downsample_bwd": (I, [P, P, P, P, I, I, P]),
    "dreg_voxel_downsample_plan": (I, [P] * 11 + [Z, I, I, F, P]),
    "dreg_voxel_downsample_plan_frozen": (I, [P] * 11 + [Z, I, I, F, P, P]),
    "dreg_voxel_segment_mean": (I, [P] * 5 + [I, I, P]),
    "dreg_grad_norm": (I, [P, P, P, Z, P]),
    "dreg_adamw_step": (I, [P] * 5 + [Z] + [F] * 5 + [I, F, P]),
    # ngp.hip
    "dreg_ngp_level_table": (ctypes.c_uint32, [F, I, I, P, P, P, P, P]),
    "dreg_f32_to_f16": (I, [P, P, Z, P]),
    "dreg_ngp_density_fwd": (I, [P] * 6 + [P] * 5 + [P, I, P]),
    "dreg_ngp_rgb_mean_fwd": (I, [P] * 6 + [I, I, P]),
    "dreg_ngp_dir_bias": (I, [P, P, P, I, P]),
    "dreg_ngp_alpha_keep": (I, [P, P, P, I, F, F, P]),
    "dreg_ngp_density_fwd_contract": (I, [P] * 6 + [P] * 5 + [P, I, I, P]),
    "dreg_ngp_density_fwd_ws": (I, [P] * 6 + [P] * 5 + [P, I, I, P, Z, P, I, P]),
    "dreg_grid_sample_points_ordered": (I, [P, P, P, P, P, I, I, I, P, I, P]),
    "dreg_grid_x_order_workspace_bytes": (Z, [I, I, I]),
    "dreg_grid_x_order": (I, [P, P, P, P, Z, I, I, I, I, P]),
    "dreg_grid_occupied_workspace_bytes": (Z, [I, I, I]),
    "dreg_grid_occupied_totals": (P, [P, I, I, I]),
    "dreg_grid_occupied_count": (I, [P, P, Z, I, I, I, P]),
    "dreg_grid_occupied_build": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "dreg_ngp_density_keep_fwd_ws": (I, [P] * 6 + [P] * 5 + [P, I, I, P, Z, P, I] + [P, P, F, F, P]),
    "dreg_grid_write_kept": (I, [P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "dreg_ngp_density_workspace_bytes": (Z, [I]),
    "dreg_ngp_rgb_dir_fwd": (I, [P] * 6 + [I, P]),
    "dreg_grid_scatter7": (I, [P] * 6 + [I, P]),
    "dreg_grid_sample_points": (I, [P, P, P, I, I, I, P, I, P]),
    # visibility.hip
    "dreg_surface_visibility": (I, [P] * 7 + [P] * 5 + [P, P, P] + [I] * 5 + [F, F, F, F, P]),
    "dreg_surface_visibility_queue": (I, [P] * 7 + [P] * 5 + [P, P, P] + [I] * 5 + [F, F, F, F, P, P, P]),
    "dreg_occupancy_coarse_bits": (I, [P, P, I, I, I, P]),
    "dreg_surface_visibility_desc_bytes": (Z, []),
    "dreg_surface_visibility_fill_desc": (I, [P] + [P] * 7 + [P] * 5 + [P, P, P] + [I] * 5 + [F, F, F, F, P, P]),
    "dreg_surface_visibility_multi": (I, [P, I, ctypes.c_long, P]),
    "dreg_surface_visibility_multi_waves": (I, [P, I, ctypes.c_long, I, P]),
}


# include/dreg_nerf_probe.h: process-global kernel-variant setters, exported by the MEASUREMENT build only (libdreg_nerf_hip_probe.so)
PROBE_SIGNATURES = {
    "dreg_conv1_bnrelu_a_probe": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "dreg_conv_set_glds": (None, [I]),
    "dreg_conv_set_wgrad_splits": (None, [I]),
    "dreg_conv_set_wgrad_big": (None, [I]),
    "dreg_conv_igemm_probe": (None, [I]),
    "dreg_conv_set_igemm_ap": (None, [I]),
    "dreg_conv_set_igemm_ap256": (None, [I]),
    "dreg_conv_set_pointwise_rmw_cin": (None, [I]),
    "dreg_conv_igemm_probe_read": (I, [P]),
    "dreg_conv_set_wgrad_pipe": (None, [I]),
    "dreg_conv_wgrad_probe_read": (I, [P]),
    "dreg_conv_set_wgrad_ring": (None, [I]),
    "dreg_conv_set_wgrad_rows_fast": (None, [I]),
    "dreg_conv_set_row_splits": (None, [I]),
    "dreg_conv_set_glds_stages": (None, [I]),
    "dreg_conv_set_wgrad_target_blocks": (None, [I]),
    "dreg_conv_set_narrow_small": (None, [I]),
    "dreg_bn_set_debug_skip": (None, [I]),
    "dreg_bn_set_store_g": (None, [I]),
    "dreg_bn_set_small_regs": (None, [I]),
    "dreg_sstem_set_pool_blocks": (None, [I]),
    "dreg_voxel_set_own_sort": (None, [I]),
    "dreg_conv_set_bn_stats_epilogue": (None, [I]),
    "dreg_conv3_halo_set_variant": (None, [I]),
    "dreg_conv3_halo64_set": (None, [I]),
    "dreg_conv3_halo_set_prof": (None, [P]),
    "dreg_bn_set_small_max_voxels": (None, [I]),
    "dreg_ngp_set_rgb_chunks": (None, [I]),
    "dreg_ngp_set_density_unroll": (None, [I]),
    "dreg_ngp_set_xcd_levels": (None, [I]),
    "dreg_visibility_set_waves": (None, [I]),
    "dreg_visibility_set_pass_bound": (I, [ctypes.c_long]),
}
PROBE_LIB_PATH = os.path.join(_HERE, "libdreg_nerf_hip_probe.so")
_probe_lib = None


class ExecOpts(ctypes.Structure):
    """dreg_exec_opts of include/dreg_nerf.h (creation options of one trunk executor)."""
    _fields_ = [(n, c_int) for n in ("sparse_grads", "bn_batch_tails", "fuse_stem", "sparse_stem", "fold_res_bn", "fold_splitk", "group_wgrad",
                                     "s2_accumulate", "fuse_bn_stats", "brick", "defer_head_pg", "persistent_deep", "guard")] + [("reserved", c_int * 3)]


def load_probe():
    """The measurement build (every entry point of the product + the setters of include/dreg_nerf_probe.h).  Tools and variant tests only."""
    global _probe_lib
    if _probe_lib is None:
        if not os.path.exists(PROBE_LIB_PATH):
            raise DregError(f"{PROBE_LIB_PATH} not found: build it with `python -m dreg_nerf_amd.build`")
        lib = ctypes.CDLL(PROBE_LIB_PATH)
        for name, (rt, at) in list(SIGNATURES.items()) + list(PROBE_SIGNATURES.items()):
            _sig(lib, name, rt, at)
        _probe_lib = lib
    return _probe_lib


def use_probe():
    """Stand-alone tools only: the measurement build for the rest of the process (every L.load() returns it from now on)."""
    global _lib
    _lib = load_probe()
    return _lib


class probe:
    """``with L.probe() as lib:`` — inside the block every ``L.load()`` of the package returns the measurement build, so the wrappers of
    dreg_nerf_amd run its kernels and ``lib.dreg_*_set_*`` selects their variants.  On exit the knobs that were touched through ``set()`` are put
    back and ``L.load()`` is the product library again.  Objects that keep a handle (executors) must be created AND dropped inside the block.
    The swap is of a module global and is NOT thread-safe: do not enter / leave the block while background threads of this package (the prefetching
    loader, the label / geometry threads, the evaluation pipeline's workers) may call ``L.load()`` — they would launch on whichever build is current."""

    def __init__(self):
        self._restore = []

    def __enter__(self):
        global _lib
        self._saved = _lib
        _lib = load_probe()
        self.lib = _lib
        return self

    def set(self, name, value, default):
        getattr(self.lib, name)(value)
        self._restore.append((name, default))

    def __getattr__(self, name):
        return getattr(self.__dict__["lib"], name)

    def __exit__(self, *exc):
        global _lib
        for name, default in reversed(self._restore):
            getattr(self.lib, name)(default)
        _lib = self._saved
        return False


def load():
    """Load the library once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DregError(
                f"{LIB_PATH} not found: build it with `python -m dreg_nerf_amd.build` "
                "(hipcc --offload-arch=gfx950); dreg_nerf_amd has no CPU/eager fallback")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (rt, at) in SIGNATURES.items():
            _sig(lib, name, rt, at)
        _lib = lib
    return _lib


def declared_symbols():
    return list(SIGNATURES.keys())


def probe_symbols():
    return list(PROBE_SIGNATURES.keys())


def ptr(t):
    if t is None:
        return None
    assert t.is_contiguous(), "non-contiguous tensor handed to the C ABI"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw hipStream_t of torch's current stream on the current device (the C-level getter costs ~0.2 us; the
    torch.cuda.current_stream() object path costs several microseconds per launch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_DEBUG_SYNC = bool(int(os.environ.get("DREG_DEBUG_SYNC", "0")))


def check(rc, name):
    if rc != 0:
        raise DregError(f"{name} failed with code {rc}")
    if _DEBUG_SYNC:  # localise asynchronous faults: DREG_DEBUG_SYNC=1
        import sys
        print(f"[dreg] {name} launched", file=sys.stderr, flush=True)
        torch.cuda.synchronize()


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    raise DregError(f"unsupported dtype {t.dtype}")


def torch_dtype(dt: int):
    return torch.bfloat16 if dt == DT_BF16 else torch.float32


def to_device_async(data, dtype, device):
    """Small host list -> device tensor without stalling the stream: pinned staging + non-blocking copy.  (On ROCm a copy from
    pageable memory — torch.tensor(list, device=...) — blocks the host until ALL previously queued work has finished.)"""
    t = torch.tensor(data, dtype=dtype)
    if device.type != "cuda":
        return t
    return t.pin_memory().to(device, non_blocking=True)
