"""ctypes binding of libdnagpu.so (include/dnagpu.h, include/dnaadjust_c.h).

The product path is the HIP library: importing this module fails loudly when the
shared object has not been built (python __graft_entry__.py build).  There is no
CPU fallback anywhere in the package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdnagpu.so")


class DnaGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dnagpu error {code}: {msg}")
        self.code = code


DNAGPU_OK = 0
DNAGPU_EINVAL = -1
DNAGPU_ENOMEM = -2
DNAGPU_EHIP = -3
DNAGPU_ENOTPOSDEF = -4
DNAGPU_ENODEVICE = -5

_lib = None


class DnaAdjSettings(C.Structure):
    """dnaadj_settings (include/dnaadjust_c.h)"""
    _fields_ = [("bst_file", C.c_char_p), ("bms_file", C.c_char_p), ("asl_file", C.c_char_p), ("seg_file", C.c_char_p),
                ("adjust_mode", C.c_int), ("multi_thread", C.c_int), ("max_iterations", C.c_int),
                ("iteration_threshold", C.c_float), ("free_std_dev", C.c_double), ("fixed_std_dev", C.c_double),
                ("scale_normals_to_unity", C.c_int), ("device", C.c_int), ("confidence_interval", C.c_float),
                ("output_tstat", C.c_int), ("network_name", C.c_char_p), ("output_folder", C.c_char_p), ("reuse_inverses", C.c_int), ("schur_carry", C.c_int), ("stage", C.c_int), ("keep_factors", C.c_int),
                ("dist_rank", C.c_int), ("dist_world", C.c_int), ("n_devices", C.c_int), ("devices", C.POINTER(C.c_int)),
                ("dist_transport", C.c_char_p), ("dist_two_level", C.c_int), ("defer_variances", C.c_int), ("batch_blocks", C.c_int), ("reuse_factors", C.c_int), ("chain_runs", C.c_int)]


class DnaAdjStatistics(C.Structure):
    """dnaadj_statistics (include/dnaadjust_c.h)"""
    _fields_ = [("chi_squared", C.c_double), ("sigma_zero", C.c_double), ("global_pelzer", C.c_double),
                ("chi_upper_limit", C.c_double), ("chi_lower_limit", C.c_double), ("measurement_params", C.c_uint32),
                ("unknown_params", C.c_uint32), ("potential_outliers", C.c_uint32), ("test_result", C.c_uint32),
                ("degrees_of_freedom", C.c_int)]


class DnaAdjInstanceStats(C.Structure):
    """dnaadj_instance_stats (include/dnaadjust_c.h)"""
    _fields_ = [("rank", C.c_int), ("device", C.c_int), ("rccl_ranks", C.c_int), ("solves", C.c_uint32), ("eliminations", C.c_uint32),
                ("completions", C.c_uint32), ("algorithmic_flops", C.c_double), ("solve_flops", C.c_double), ("exchanged_bytes", C.c_uint64),
                ("exchange_ms", C.c_double), ("chain_ms", C.c_double)]


class DnaSynthSpec(C.Structure):
    _fields_ = [("rows", C.c_uint32), ("cols", C.c_uint32), ("n_baselines", C.c_uint64), ("n_blocks", C.c_uint32),
                ("seed", C.c_uint64), ("initial_sigma", C.c_double), ("x_clusters", C.c_uint32), ("y_cluster", C.c_uint32),
                ("y_llh", C.c_uint32), ("scalars", C.c_uint32), ("rows_lo", C.c_uint32), ("rows_hi", C.c_uint32), ("ragged", C.c_double)]


class DnaImportSummary(C.Structure):
    _fields_ = [("stations", C.c_uint64), ("records", C.c_uint64), ("vectors", C.c_uint64), ("clusters", C.c_uint64),
                ("vectors_transformed", C.c_uint64)]


class DnaSynthSummary(C.Structure):
    _fields_ = [("stations", C.c_uint64), ("baselines", C.c_uint64), ("measurement_rows", C.c_uint64), ("blocks", C.c_uint64),
                ("max_block_unknowns", C.c_uint64)]

c_u32p = C.POINTER(C.c_uint32)
c_f64p = C.POINTER(C.c_double)


def _sig(lib, name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes
    return f


def load():
    """Load libdnagpu.so and declare every exported prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()')")
    # kernel-tuning experiments (tools/build_variant.sh) load an alternative build of the same library
    # (only effective if the HIP runtime has not started in this process yet: see bench.py / INTEGRATION.md)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    lib = C.CDLL(os.environ.get("DNAGPU_LIB_OVERRIDE") or LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    i = C.c_int
    u32 = C.c_uint32
    sz = C.c_size_t
    _sig(lib, "dnagpu_device_count", i, [])
    _sig(lib, "dnagpu_create", i, [i, C.POINTER(vp)])
    _sig(lib, "dnagpu_destroy", None, [vp])
    _sig(lib, "dnagpu_last_error", C.c_char_p, [vp])
    _sig(lib, "dnagpu_last_info", i, [vp])
    _sig(lib, "dnagpu_sync", i, [vp])
    _sig(lib, "dnagpu_cholesky_inverse_packed", i, [vp, c_f64p, u32, i])
    _sig(lib, "dnagpu_multiply_sym_packed", i, [vp, c_f64p, c_f64p, c_f64p, u32])
    _sig(lib, "dnagpu_profile_enable", i, [vp, i])
    _sig(lib, "dnagpu_debug_fail_allocation", i, [C.c_long])
    _sig(lib, "dnagpu_debug_fail_batch_workspaces", i, [C.c_long])
    _sig(lib, "dnagpu_debug_set_small_tiles", C.c_long, [C.c_long])
    _sig(lib, "dnagpu_debug_set_tiny_tiles", C.c_long, [C.c_long])
    _sig(lib, "dnagpu_debug_set_info_carry", i, [i])
    _sig(lib, "dnagpu_info_carry", i, [vp])
    _sig(lib, "dnagpu_ctx_set_info_carry", i, [vp, i])
    _sig(lib, "dnagpu_debug_tile_order", C.c_long, [C.c_int] * 8 + [C.POINTER(C.c_uint32), C.c_long, C.POINTER(C.c_int)])
    _sig(lib, "dnagpu_profile_reset", i, [vp])
    _sig(lib, "dnagpu_profile_get", i, [vp, c_f64p, c_f64p, C.POINTER(C.c_uint64)])
    _sig(lib, "dnagpu_matrix_create", i, [vp, u32, C.POINTER(vp)])
    _sig(lib, "dnagpu_matrix_destroy", None, [vp, vp])
    _sig(lib, "dnagpu_matrix_reset", i, [vp, i, vp, u32])
    _sig(lib, "dnagpu_matrix_upload_packed", i, [vp, i, vp, c_f64p, u32])
    _sig(lib, "dnagpu_matrix_download_packed", i, [vp, i, vp, c_f64p])
    _sig(lib, "dnagpu_matrix_download_packed_async", i, [vp, i, vp, c_f64p])
    _sig(lib, "dnagpu_copies_sync", i, [vp])
    _sig(lib, "dnagpu_matrix_copy", i, [vp, i, vp, vp])
    _sig(lib, "dnagpu_matrix_export", i, [vp, i, vp, vp, sz])
    _sig(lib, "dnagpu_matrix_import", i, [vp, i, vp, vp, u32])
    _sig(lib, "dnagpu_invert", i, [vp, i, vp, i])
    _sig(lib, "dnagpu_block_create", i, [vp, u32, u32, u32])
    _sig(lib, "dnagpu_block_destroy", i, [vp, u32])
    _sig(lib, "dnagpu_block_set_stations", i, [vp, u32, c_f64p])
    _sig(lib, "dnagpu_block_reset_stations", i, [vp, i, u32, vp, i])
    _sig(lib, "dnagpu_chain_hold_info", i, [vp, i, i])
    _sig(lib, "dnagpu_chain_take_info", i, [vp, i])
    _sig(lib, "dnagpu_block_table_create", i, [vp, u32, vp, vp, vp, vp, C.POINTER(vp)])
    _sig(lib, "dnagpu_block_table_apply", i, [vp, i, vp, i, i])
    _sig(lib, "dnagpu_block_table_destroy", None, [vp, vp])
    _sig(lib, "dnagpu_block_set_baselines", i, [vp, u32, c_u32p, c_u32p, c_f64p, c_f64p])
    _sig(lib, "dnagpu_block_set_clusters", i, [vp, u32, c_u32p, c_u32p, c_f64p, u32, c_u32p, c_f64p])
    _sig(lib, "dnagpu_block_get_stations", i, [vp, i, u32, i, c_f64p])
    _sig(lib, "dnagpu_block_put_stations", i, [vp, i, u32, i, c_f64p])
    _sig(lib, "dnagpu_block_copy_stations", i, [vp, i, u32, i, i])
    _sig(lib, "dnagpu_block_compute_b", i, [vp, i, u32])
    _sig(lib, "dnagpu_block_get_b", i, [vp, i, u32, c_f64p])
    _sig(lib, "dnagpu_block_get_weights", i, [vp, i, u32, c_f64p])
    _sig(lib, "dnagpu_block_msr_statistics", i, [vp, i, u32, vp, c_f64p, c_f64p])
    _sig(lib, "dnagpu_block_set_station_geo", i, [vp, u32, c_f64p, c_f64p, c_f64p])
    _sig(lib, "dnagpu_block_set_terrestrial", i, [vp, u32, u32, C.c_char_p, c_u32p, c_f64p, c_f64p, c_f64p, c_f64p, c_f64p, c_u32p, c_u32p, u32])
    _sig(lib, "dnagpu_block_set_direction_sets", i, [vp, u32, u32, c_u32p, c_u32p, c_f64p])
    _sig(lib, "dnagpu_block_update_geodetic", i, [vp, i, u32])
    _sig(lib, "dnagpu_block_get_station_llh", i, [vp, i, u32, c_f64p])
    _sig(lib, "dnagpu_block_get_terrestrial", i, [vp, i, u32, c_f64p, c_f64p])
    _sig(lib, "dnagpu_block_terrestrial_precisions", i, [vp, i, u32, vp, c_f64p])
    _sig(lib, "dnagpu_form_normals", i, [vp, i, u32, vp])
    _sig(lib, "dnagpu_add_diag3x3", i, [vp, i, vp, c_u32p, c_f64p, sz, i])
    _sig(lib, "dnagpu_form_rhs", i, [vp, i, u32])
    _sig(lib, "dnagpu_solve_corrections", i, [vp, i, u32, vp])
    _sig(lib, "dnagpu_update_estimates", i, [vp, i, u32, c_f64p, c_u32p])
    _sig(lib, "dnagpu_block_get_corrections", i, [vp, i, u32, c_f64p])
    _sig(lib, "dnagpu_block_get_rhs", i, [vp, i, u32, c_f64p])
    _sig(lib, "dnagpu_junction_gather", i, [vp, i, u32, vp, c_u32p, sz, vp])
    _sig(lib, "dnagpu_schur_carry", i, [vp, i, u32, vp, c_u32p, sz, vp])
    _sig(lib, "dnagpu_junction_export", i, [vp, i, vp, vp, sz])
    _sig(lib, "dnagpu_junction_import", i, [vp, i, vp, vp, u32])
    _sig(lib, "dnagpu_junction_device_pointers", i, [vp, vp, i, vp, vp, vp, vp, vp])
    _sig(lib, "dnagpu_small_batch_create", i, [vp, u32, c_u32p, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp])
    _sig(lib, "dnagpu_small_batch_condense", i, [vp, i, vp])
    _sig(lib, "dnagpu_small_batch_solve", i, [vp, i, vp, c_f64p])
    _sig(lib, "dnagpu_small_batch_destroy", None, [vp, vp])
    _sig(lib, "dnagpu_chain_plan_create", i, [vp, sz, vp, sz, c_u32p, C.c_double, vp])
    _sig(lib, "dnagpu_chain_plan_run", i, [vp, i, vp, sz])
    _sig(lib, "dnagpu_chain_plan_run_rhs", i, [vp, i, vp, sz, sz])
    _sig(lib, "dnagpu_chain_plan_destroy", None, [vp, vp])
    _sig(lib, "dnagpu_chain_step_rhs", i, [vp, i, u32, u32, c_u32p, sz, vp, vp, c_u32p, sz, vp, c_u32p, sz, vp])
    _sig(lib, "dnagpu_schur_carry_keep", i, [vp, i, u32, vp, c_u32p, sz, vp, vp])
    _sig(lib, "dnagpu_schur_carry_rhs", i, [vp, i, u32, c_u32p, sz, vp, vp])
    _sig(lib, "dnagpu_block_reduce", i, [vp, i, u32, vp, c_u32p, sz, vp, vp])
    _sig(lib, "dnagpu_mem_info", i, [vp, C.POINTER(sz), C.POINTER(sz)])
    _sig(lib, "dnagpu_device_alloc", i, [vp, sz, C.POINTER(vp)])
    _sig(lib, "dnagpu_device_free", None, [vp, vp])
    _sig(lib, "dnagpu_copy", i, [vp, vp, vp, sz])
    _sig(lib, "dnagpu_matrix_resize", i, [vp, vp, u32])
    _sig(lib, "dnagpu_set_inverse_exchange", i, [vp, i, i, vp, vp])
    _sig(lib, "dnagpu_inverse_exchange_stats", i, [vp, C.POINTER(C.c_uint64), c_f64p])
    _sig(lib, "dnagpu_matrix_device_pointers", i, [vp, C.POINTER(c_f64p), C.POINTER(c_f64p), c_u32p])
    _sig(lib, "dnagpu_host_alloc", i, [vp, sz, C.POINTER(vp)])
    _sig(lib, "dnagpu_host_free", None, [vp, vp])
    _sig(lib, "dnagpu_block_form_reduce", i, [vp, i, u32, c_u32p, c_f64p, sz, c_u32p, sz, vp, vp])
    _sig(lib, "dnagpu_batch_reserve", i, [vp, i, u32, u32, i, C.POINTER(C.c_int)])
    _sig(lib, "dnagpu_block_form_reduce_batched", i, [vp, i, i, c_u32p, vp, vp, C.POINTER(C.c_size_t), vp, C.POINTER(C.c_size_t), vp, vp, C.POINTER(C.c_int)])
    _sig(lib, "dnagpu_partial_complete_factor_batched", i, [vp, i, i, vp, vp, C.POINTER(C.c_int)])
    _sig(lib, "dnagpu_partial_finish_batched", i, [vp, i, i, vp, vp])
    _sig(lib, "dnagpu_partial_create", i, [vp, u32, u32, C.POINTER(vp)])
    _sig(lib, "dnagpu_partial_create_in", i, [vp, u32, u32, vp, C.POINTER(vp)])
    _sig(lib, "dnagpu_partial_create_spine", i, [vp, u32, u32, vp, C.POINTER(vp)])
    _sig(lib, "dnagpu_partial_destroy", None, [vp, vp])
    _sig(lib, "dnagpu_partial_complete", i, [vp, i, vp, vp, vp])
    _sig(lib, "dnagpu_partial_reduce_rhs", i, [vp, i, u32, vp, vp])
    _sig(lib, "dnagpu_partial_complete_factor", i, [vp, i, vp, vp])
    _sig(lib, "dnagpu_partial_solve", i, [vp, i, u32, vp])
    _sig(lib, "dnagpu_partial_finish", i, [vp, i, vp, vp])
    _sig(lib, "dnagpu_block_load_reduced", i, [vp, i, u32, u32, c_u32p, sz, vp, vp])
    _sig(lib, "dnagpu_junction_scatter", i, [vp, i, vp, c_u32p, sz, vp])
    _sig(lib, "dnagpu_block_add_rhs", i, [vp, i, u32, c_u32p, sz, vp, i])
    _sig(lib, "dnagpu_block_gather_stations", i, [vp, i, u32, c_u32p, u32, c_u32p, sz])
    _sig(lib, "dnagpu_junction_rhs", i, [vp, i, u32, c_u32p, sz, vp])
    _sig(lib, "dnagpu_junction_get_estimates", i, [vp, i, vp, c_f64p])
    _sig(lib, "dnagpu_junction_put_estimates", i, [vp, i, vp, c_f64p, sz])
    _sig(lib, "dnagpu_chain_wait", i, [vp, i, i])
    _sig(lib, "dnagpu_chain_sync", i, [vp, i])
    # ---- include/dnaadjust_c.h ------------------------------------------------
    u64 = C.c_uint64
    _sig(lib, "dnaadj_default_settings", None, [C.POINTER(DnaAdjSettings)])
    _sig(lib, "dnaadj_create", i, [C.POINTER(vp)])
    _sig(lib, "dnaadj_destroy", None, [vp])
    _sig(lib, "dnaadj_last_error", C.c_char_p, [vp])
    _sig(lib, "dnaadj_prepare", i, [vp, C.POINTER(DnaAdjSettings)])
    _sig(lib, "dnaadj_adjust", i, [vp, C.POINTER(i)])
    _sig(lib, "dnaadj_cancel", i, [vp])
    _sig(lib, "dnaadj_reset", i, [vp])
    _sig(lib, "dnaadj_block_count", u32, [vp])
    _sig(lib, "dnaadj_iterations", u32, [vp])
    _sig(lib, "dnaadj_max_correction", C.c_double, [vp])
    _sig(lib, "dnaadj_iteration_correction", C.c_double, [vp, u32])
    _sig(lib, "dnaadj_measurement_count", u32, [vp])
    _sig(lib, "dnaadj_unknowns_count", u32, [vp])
    _sig(lib, "dnaadj_degrees_of_freedom", i, [vp])
    _sig(lib, "dnaadj_adjust_time_ms", C.c_double, [vp])
    _sig(lib, "dnaadj_solve_flops", C.c_double, [vp])
    _sig(lib, "dnaadj_solve_count", u32, [vp])
    _sig(lib, "dnaadj_elimination_count", u32, [vp])
    _sig(lib, "dnaadj_completion_count", u32, [vp])
    _sig(lib, "dnaadj_factor_reuses", C.c_uint64, [vp])
    _sig(lib, "dnaadj_minimal_work_flops", C.c_double, [vp])
    _sig(lib, "dnaadj_chain_step_reuses", C.c_uint64, [vp])
    _sig(lib, "dnaadj_small_batch_steps", C.c_uint64, [vp])
    _sig(lib, "dnaadj_chain_runs", C.c_int, [vp])
    _sig(lib, "dnaadj_algorithmic_flops", C.c_double, [vp])
    _sig(lib, "dnaadj_station_count", u32, [vp])
    _sig(lib, "dnaadj_block_station_count", u32, [vp, u32])
    _sig(lib, "dnaadj_block_stations", i, [vp, u32, c_u32p])
    _sig(lib, "dnaadj_block_estimates", i, [vp, u32, c_f64p])
    _sig(lib, "dnaadj_block_variances_packed", i, [vp, u32, c_f64p])
    _sig(lib, "dnaadj_adjusted_coordinates", i, [vp, c_f64p])
    _sig(lib, "dnaadj_device_context", vp, [vp])
    _sig(lib, "dnaimport_text", i, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(DnaImportSummary), C.c_char_p, sz])
    _sig(lib, "dnaimport_text_geo", i, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(DnaImportSummary), C.c_char_p, sz])
    _sig(lib, "dnaadj_dist_rccl_available", i, [])
    _sig(lib, "dnaadj_dist_unique_id", i, [C.c_char_p, C.c_char_p, sz])
    _sig(lib, "dnaadj_dist_attach_rccl", i, [vp, i, i, C.c_char_p, i])
    _sig(lib, "dnaadj_adjust_distributed", i, [vp, C.POINTER(i)])
    _sig(lib, "dnaadj_dist_info", i, [vp, C.POINTER(i), C.POINTER(i), C.c_char_p, sz])
    _sig(lib, "dnaadj_block_owner", i, [vp, u32])
    _sig(lib, "dnaadj_exchange_stats", i, [vp, C.POINTER(C.c_uint64), c_f64p, c_f64p])
    _sig(lib, "dnaadj_device_instance_context", vp, [vp, i])
    _sig(lib, "dnaadj_device_instance_stats", i, [vp, i, C.POINTER(DnaAdjInstanceStats)])
    _sig(lib, "dnaadj_debug_cancel_instance", i, [vp, i])
    _sig(lib, "dnaadj_debug_tcp_share_unique_id", i, [i, i, C.c_char_p, C.c_char_p, i, C.c_double, C.c_char_p, sz])
    _sig(lib, "dnaadj_generate_statistics", i, [vp])
    _sig(lib, "dnaadj_get_statistics", i, [vp, C.POINTER(DnaAdjStatistics)])
    _sig(lib, "dnaadj_measurement_record_count", u64, [vp])
    _sig(lib, "dnaadj_measurement_records", i, [vp, vp, u64])
    _sig(lib, "dnaadj_block_prec_adj_msrs_count", u64, [vp, u32])
    _sig(lib, "dnaadj_block_prec_adj_msrs", i, [vp, u32, c_f64p, u64])
    _sig(lib, "dnaadj_serialise_adjusted_variance_matrices", i, [vp])
    _sig(lib, "dnaadj_deserialise_adjusted_variance_matrices", i, [vp])
    _sig(lib, "dnaadj_update_binary_files", i, [vp])
    _sig(lib, "dnastat_normal_quantile", C.c_double, [C.c_double])
    _sig(lib, "dnastat_chi_squared_quantile", C.c_double, [C.c_double, C.c_double])
    ip = C.POINTER(C.c_int)
    dp = C.POINTER(C.c_double)
    _sig(lib, "dnaadj_block_flags", i, [vp, u32, ip, ip, ip])
    _sig(lib, "dnaadj_junction_unknowns", u32, [vp, u32])
    _sig(lib, "dnaadj_junction_payload_doubles", sz, [vp, u32])
    _sig(lib, "dnaadj_phased_begin_iteration", i, [vp])
    _sig(lib, "dnaadj_phased_forward_block", i, [vp, u32, dp])
    _sig(lib, "dnaadj_phased_reverse_block", i, [vp, u32, dp])
    _sig(lib, "dnaadj_phased_combine_block", i, [vp, u32, dp])
    _sig(lib, "dnaadj_phased_finalise_block", i, [vp, u32])
    _sig(lib, "dnaadj_phased_note_correction", i, [vp, C.c_double])
    _sig(lib, "dnaadj_phased_end_iteration", i, [vp, ip])
    _sig(lib, "dnaadj_phased_finish", i, [vp, ip])
    _sig(lib, "dnaadj_staged", i, [vp])
    _sig(lib, "dnagpu_profile_hbm_enable", i, [vp, i])
    _sig(lib, "dnagpu_profile_hbm_get", i, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), i])
    _sig(lib, "dnaadj_oscillation_history", sz, [vp, C.POINTER(C.c_double), sz])
    _sig(lib, "dnaadj_summaries", sz, [vp, sz, C.c_char_p, sz])
    _sig(lib, "dnaadj_plan_distributed", i, [vp, C.POINTER(DnaAdjSettings), i, C.c_double, C.c_char_p, sz, C.POINTER(sz)])
    _sig(lib, "dnaadj_memory_plan", i, [vp, C.POINTER(C.c_double)])
    _sig(lib, "dnaadj_dist_set_timeout", None, [C.c_double])
    _sig(lib, "dnaadj_debug_stall_rank", None, [C.c_int, C.c_long, C.c_double])
    _sig(lib, "dnaadj_condensed_schedule", i, [vp])
    _sig(lib, "dnaadj_batched_block_steps", C.c_uint64, [vp])
    _sig(lib, "dnaadj_batched_flops", C.c_double, [vp])
    _sig(lib, "dnaadj_condensed_payload_doubles", sz, [vp, u32])
    _sig(lib, "dnaadj_phased_condense_block", i, [vp, u32])
    _sig(lib, "dnaadj_phased_condensed_forward", i, [vp, u32])
    _sig(lib, "dnaadj_phased_condensed_reverse", i, [vp, u32])
    _sig(lib, "dnaadj_phased_rigorous_block", i, [vp, u32, C.POINTER(C.c_double)])
    _sig(lib, "dnaadj_phased_condense_blocks", i, [vp, c_u32p, sz])
    _sig(lib, "dnaadj_phased_condensed_chains", i, [vp])
    _sig(lib, "dnaadj_phased_rigorous_blocks", i, [vp, c_u32p, sz])
    _sig(lib, "dnaadj_condensed_export", i, [vp, u32, vp])
    _sig(lib, "dnaadj_condensed_import", i, [vp, u32, vp])
    _sig(lib, "dnaadj_statistics_prepare", i, [vp])
    _sig(lib, "dnaadj_statistics_blocks", i, [vp, c_u32p, sz])
    _sig(lib, "dnaadj_statistics_get_partial", i, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)])
    _sig(lib, "dnaadj_statistics_set_partial", i, [vp, C.c_double, u32])
    _sig(lib, "dnaadj_record_statistics_get", i, [vp, c_f64p, C.c_uint64])
    _sig(lib, "dnaadj_record_statistics_set", i, [vp, c_f64p, C.c_uint64])
    _sig(lib, "dnaadj_statistics_finish", i, [vp])
    _sig(lib, "dnaadj_junction_export", i, [vp, i, u32, vp])
    _sig(lib, "dnaadj_junction_import", i, [vp, i, u32, vp])
    _sig(lib, "dnaadj_block_get_coords", i, [vp, u32, i, c_f64p])
    _sig(lib, "dnaadj_block_set_coords", i, [vp, u32, c_f64p])
    _sig(lib, "dnaadj_block_recompute_b", i, [vp, u32])
    _sig(lib, "dnasynth_write_network", i, [C.c_char_p, C.c_char_p, C.POINTER(DnaSynthSpec), C.POINTER(DnaSynthSummary), C.c_char_p, sz])
    _sig(lib, "dnaio_file_summary", i, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64), C.c_char_p, sz])
    _sig(lib, "dnaio_seg_summary", i, [C.c_char_p, C.c_char_p, c_u32p, c_u32p, u32, C.c_char_p, sz])
    _sig(lib, "dnaio_sizeof_station", sz, [])
    _sig(lib, "dnaio_sizeof_measurement", sz, [])
    _lib = lib
    return lib


EXPORTED_DNAGPU = [
    "dnagpu_device_count", "dnagpu_create", "dnagpu_destroy", "dnagpu_last_error", "dnagpu_last_info", "dnagpu_sync",
    "dnagpu_cholesky_inverse_packed", "dnagpu_multiply_sym_packed", "dnagpu_profile_enable", "dnagpu_profile_reset", "dnagpu_debug_fail_allocation", "dnagpu_debug_fail_batch_workspaces", "dnagpu_debug_set_small_tiles", "dnagpu_debug_set_tiny_tiles", "dnagpu_debug_set_info_carry", "dnagpu_info_carry", "dnagpu_ctx_set_info_carry", "dnagpu_chain_reserve", "dnagpu_copy_stage_reserve", "dnagpu_partial_pack_device", "dnagpu_partial_unpack_device", "dnagpu_partial_pack_host_async", "dnagpu_partial_unpack_host", "dnagpu_debug_tile_order",
    "dnagpu_block_keep_corrections", "dnagpu_osc_reset", "dnagpu_osc_block", "dnagpu_osc_blocks", "dnagpu_form_rhs_batched", "dnagpu_osc_flagged", "dnagpu_osc_block_visits", "dnagpu_profile_get", "dnagpu_profile_hbm_enable", "dnagpu_profile_hbm_get", "dnagpu_matrix_pack_device", "dnagpu_matrix_unpack_device", "dnagpu_matrix_create", "dnagpu_matrix_destroy", "dnagpu_matrix_reset",
    "dnagpu_matrix_upload_packed", "dnagpu_matrix_download_packed", "dnagpu_matrix_download_packed_async", "dnagpu_copies_sync", "dnagpu_matrix_copy", "dnagpu_matrix_export", "dnagpu_matrix_import", "dnagpu_invert",
    "dnagpu_block_create", "dnagpu_block_destroy", "dnagpu_block_set_stations", "dnagpu_block_reset_stations", "dnagpu_chain_hold_info", "dnagpu_chain_take_info", "dnagpu_block_table_create", "dnagpu_block_table_apply", "dnagpu_block_table_destroy", "dnagpu_block_set_baselines", "dnagpu_block_set_clusters",
    "dnagpu_block_get_stations", "dnagpu_block_put_stations", "dnagpu_block_copy_stations", "dnagpu_block_compute_b",
    "dnagpu_block_get_b", "dnagpu_block_get_weights", "dnagpu_block_msr_statistics", "dnagpu_block_set_station_geo", "dnagpu_block_set_terrestrial",
    "dnagpu_block_set_direction_sets", "dnagpu_block_update_geodetic", "dnagpu_block_get_station_llh", "dnagpu_block_get_terrestrial", "dnagpu_block_terrestrial_precisions", "dnagpu_form_normals", "dnagpu_add_diag3x3", "dnagpu_form_rhs",
    "dnagpu_solve_corrections", "dnagpu_update_estimates", "dnagpu_block_get_corrections", "dnagpu_block_get_rhs",
    "dnagpu_block_add_rhs", "dnagpu_block_gather_stations", "dnagpu_junction_gather", "dnagpu_schur_carry", "dnagpu_schur_carry_keep", "dnagpu_schur_carry_rhs", "dnagpu_chain_step_rhs", "dnagpu_small_batch_create", "dnagpu_small_batch_condense", "dnagpu_small_batch_solve", "dnagpu_small_batch_destroy", "dnagpu_junction_export", "dnagpu_junction_import", "dnagpu_junction_device_pointers", "dnagpu_block_reduce", "dnagpu_block_form_reduce", "dnagpu_batch_reserve", "dnagpu_block_form_reduce_batched", "dnagpu_partial_complete_factor_batched", "dnagpu_partial_finish_batched", "dnagpu_mem_info", "dnagpu_device_alloc", "dnagpu_device_free", "dnagpu_copy", "dnagpu_matrix_resize", "dnagpu_matrix_device_pointers", "dnagpu_set_inverse_exchange", "dnagpu_inverse_exchange_stats", "dnagpu_host_alloc", "dnagpu_host_free", "dnagpu_partial_create", "dnagpu_partial_create_in", "dnagpu_partial_create_spine", "dnagpu_partial_destroy", "dnagpu_partial_complete", "dnagpu_partial_complete_factor", "dnagpu_partial_solve", "dnagpu_partial_finish", "dnagpu_partial_reduce_rhs",
    "dnagpu_block_load_reduced", "dnagpu_junction_scatter", "dnagpu_junction_rhs", "dnagpu_junction_get_estimates",
    "dnagpu_junction_put_estimates", "dnagpu_chain_wait", "dnagpu_chain_sync",
    "dnagpu_chain_plan_create", "dnagpu_chain_plan_info", "dnagpu_chain_plan_run", "dnagpu_chain_plan_run_rhs", "dnagpu_chain_plan_destroy", "dnagpu_partial_complete_factor_planned",
]

EXPORTED_DNAADJ = [
    "dnaadj_default_settings", "dnaadj_create", "dnaadj_destroy", "dnaadj_last_error", "dnaadj_prepare", "dnaadj_adjust",
    "dnaadj_cancel", "dnaadj_reset", "dnaadj_block_count", "dnaadj_iterations", "dnaadj_max_correction", "dnaadj_iteration_correction",
    "dnaadj_measurement_count", "dnaadj_unknowns_count", "dnaadj_degrees_of_freedom", "dnaadj_adjust_time_ms",
    "dnaadj_solve_flops", "dnaadj_solve_count", "dnaadj_elimination_count", "dnaadj_completion_count", "dnaadj_factor_reuses", "dnaadj_chain_step_reuses", "dnaadj_small_batch_steps", "dnaadj_chain_runs", "dnaadj_minimal_work_flops", "dnaadj_algorithmic_flops", "dnaadj_station_count", "dnaadj_block_station_count", "dnaadj_block_stations",
    "dnaadj_block_estimates", "dnaadj_block_variances_packed", "dnaadj_adjusted_coordinates", "dnaadj_device_context",
    "dnaimport_text", "dnaimport_text_geo", "dnaadj_dist_rccl_available", "dnaadj_dist_unique_id", "dnaadj_dist_attach_rccl", "dnaadj_adjust_distributed", "dnaadj_dist_info",
    "dnaadj_block_owner", "dnaadj_exchange_stats", "dnaadj_device_instance_context", "dnaadj_device_instance_stats", "dnaadj_debug_cancel_instance", "dnaadj_debug_tcp_share_unique_id",
    "dnaadj_generate_statistics", "dnaadj_get_statistics", "dnaadj_measurement_record_count", "dnaadj_measurement_records",
    "dnaadj_block_prec_adj_msrs_count", "dnaadj_block_prec_adj_msrs", "dnaadj_serialise_adjusted_variance_matrices",
    "dnaadj_deserialise_adjusted_variance_matrices", "dnaadj_update_binary_files", "dnastat_normal_quantile", "dnastat_chi_squared_quantile",
    "dnaadj_block_flags", "dnaadj_junction_unknowns", "dnaadj_junction_payload_doubles", "dnaadj_phased_begin_iteration",
    "dnaadj_phased_forward_block", "dnaadj_phased_reverse_block", "dnaadj_phased_combine_block", "dnaadj_phased_finalise_block",
    "dnaadj_phased_note_correction", "dnaadj_phased_end_iteration", "dnaadj_phased_finish", "dnaadj_staged", "dnaadj_plan_distributed", "dnaadj_oscillation_history", "dnaadj_summaries", "dnaadj_memory_plan", "dnaadj_dist_set_timeout", "dnaadj_debug_stall_rank", "dnaadj_condensed_schedule", "dnaadj_batched_block_steps", "dnaadj_batched_flops", "dnaadj_condensed_payload_doubles", "dnaadj_phased_condense_block",
    "dnaadj_phased_condensed_forward", "dnaadj_phased_condensed_reverse", "dnaadj_phased_rigorous_block", "dnaadj_phased_condense_blocks", "dnaadj_phased_condensed_chains",
    "dnaadj_phased_rigorous_blocks", "dnaadj_condensed_export",
    "dnaadj_condensed_import", "dnaadj_statistics_prepare", "dnaadj_statistics_blocks", "dnaadj_statistics_get_partial",
    "dnaadj_statistics_set_partial", "dnaadj_record_statistics_get", "dnaadj_record_statistics_set", "dnaadj_statistics_finish",
    "dnaadj_junction_export",
    "dnaadj_junction_import", "dnaadj_block_get_coords", "dnaadj_block_set_coords", "dnaadj_block_recompute_b",
    "dnasynth_write_network", "dnaio_file_summary", "dnaio_seg_summary", "dnaio_sizeof_station", "dnaio_sizeof_measurement",
]
