"""Python mirror of the reference's dna_adjust interface (dnaadjust.hpp:259-405) over the
C-ABI of include/dnaadjust_c.h.  Method names, argument meaning and error behaviour follow
the reference: PrepareAdjustment(project_settings) -> AdjustNetwork() -> getters; failures
raise NetAdjustException with the reference's message text.  The work itself runs in the
HIP library; this module contains no numerical code and no CPU fallback."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import DnaAdjInstanceStats, DnaAdjSettings, DnaAdjStatistics, DnaSynthSpec, DnaSynthSummary, c_f64p, c_u32p

SimultaneousMode = 0
PhasedMode = 1
Phased_Block_1Mode = 2

ADJUST_SUCCESS = 0
ADJUST_MAX_ITERATIONS_EXCEEDED = 1
ADJUST_THRESHOLD_EXCEEDED = 2
ADJUST_TEST_FAILED = 3
ADJUST_BLOCK_ERROR = 4
ADJUST_EXCEPTION_RAISED = 5
ADJUST_CANCELLED = 6


class NetAdjustException(RuntimeError):
    """std::runtime_error thrown by dna_adjust::SignalExceptionAdjustment (dnaadjust.cpp:10049)."""


class ProjectSettings:
    """The members of project_settings (include/config/dnaoptions.hpp) the adjustment path reads."""

    def __init__(self, network_name=None, folder=".", adjust_mode=SimultaneousMode, multi_thread=False,
                 max_iterations=10, iteration_threshold=0.0005, free_std_dev=10.0, fixed_std_dev=1e-6,
                 scale_normals_to_unity=False, device=0, confidence_interval=95.0, output_tstat=False, output_folder=None,
                 reuse_inverses=False, schur_carry=True, keep_factors=True, stage=False, dist_rank=0, dist_world=1, devices=None,
                 dist_transport=None, dist_two_level=True, defer_variances=2, batch_blocks=32, reuse_factors=True, chain_runs=-1):
        self.bst_file = self.bms_file = self.asl_file = self.seg_file = None
        self.network_name = network_name          # g.network_name
        self.output_folder = output_folder if output_folder is not None else folder   # g.output_folder
        self.confidence_interval = confidence_interval
        self.output_tstat = output_tstat          # o._adj_msr_tstat
        self.reuse_inverses = reuse_inverses      # device path only: block inverses stay resident across iterations
        self.schur_carry = schur_carry            # device path only: carry-only steps eliminate instead of inverting
        self.stage = stage                        # a.stage: rigorous variances in page-locked host memory
        self.keep_factors = keep_factors          # device path only: rigorous solves complete the condensing step's factor
        # multi-GPU: one process per GPU (dist_rank of dist_world) or one process driving `devices`
        self.dist_rank, self.dist_world, self.devices = dist_rank, dist_world, devices
        self.dist_transport, self.dist_two_level = dist_transport, dist_two_level
        self.defer_variances = defer_variances
        self.reuse_factors = reuse_factors        # GNSS-only networks: iterations >= 2 keep the factors of iteration 1 (right-hand sides only)
        self.chain_runs = chain_runs              # many small blocks on one GPU: the junction chains in this many runs advancing together (-1: choose)
        self.batch_blocks = batch_blocks          # condensed schedule: blocks of one shape as one batch of merged launches (0 / 1 = off)
        if network_name is not None:
            self.set_filenames(os.path.join(folder, network_name))
        self.adjust_mode = adjust_mode
        self.multi_thread = multi_thread
        self.max_iterations = max_iterations
        self.iteration_threshold = iteration_threshold
        self.free_std_dev = free_std_dev
        self.fixed_std_dev = fixed_std_dev
        self.scale_normals_to_unity = scale_normals_to_unity
        self.device = device

    def set_filenames(self, base):
        """adjust_settings::setFilenames (dnaoptions.hpp:455-460) + s.asl_file"""
        self.bst_file = base + ".bst"
        self.bms_file = base + ".bms"
        self.seg_file = base + ".seg"
        self.asl_file = base + ".asl"


def _b(s):
    return None if s is None else os.fsencode(s)


class DnaAdjust:
    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        if self.lib.dnaadj_create(C.byref(h)) != 0:
            raise NetAdjustException("dnaadj_create failed")
        self.h = h
        self._keep = None

    def close(self):
        if self.h:
            self.lib.dnaadj_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise NetAdjustException(self.lib.dnaadj_last_error(self.h).decode(errors="replace"))

    # ---- reference interface -------------------------------------------------
    def PrepareAdjustment(self, p):
        self._chk(self.lib.dnaadj_prepare(self.h, C.byref(self._settings(p))))

    def plan_distributed(self, p, world, hbm_bytes=309.0e9):
        """dnaadj_plan_distributed: PrepareAdjustment's plan for `world` GPUs of `hbm_bytes` each, WITHOUT a device (dict from its JSON)"""
        import json
        s = self._settings(p)
        need = C.c_size_t(0)
        self._chk(self.lib.dnaadj_plan_distributed(self.h, C.byref(s), int(world), float(hbm_bytes), None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value + 16)
        self._chk(self.lib.dnaadj_plan_distributed(self.h, C.byref(s), int(world), float(hbm_bytes), buf, need.value + 16, None))
        return json.loads(buf.value.decode())

    def _settings(self, p):
        s = DnaAdjSettings()
        self.lib.dnaadj_default_settings(C.byref(s))
        self._keep = [_b(p.bst_file), _b(p.bms_file), _b(p.asl_file), _b(p.seg_file), _b(getattr(p, "network_name", None)),
                      _b(getattr(p, "output_folder", None))]
        s.bst_file, s.bms_file, s.asl_file, s.seg_file, s.network_name, s.output_folder = self._keep
        s.confidence_interval = float(getattr(p, "confidence_interval", 95.0))
        s.output_tstat = int(bool(getattr(p, "output_tstat", False)))
        s.reuse_inverses = int(bool(getattr(p, "reuse_inverses", False)))
        s.schur_carry = int(bool(getattr(p, "schur_carry", True)))
        s.keep_factors = int(bool(getattr(p, "keep_factors", True)))
        s.stage = int(bool(getattr(p, "stage", False)))
        s.adjust_mode = int(p.adjust_mode)
        s.multi_thread = int(bool(p.multi_thread))
        s.max_iterations = int(p.max_iterations)
        s.iteration_threshold = float(p.iteration_threshold)
        s.free_std_dev = float(p.free_std_dev)
        s.fixed_std_dev = float(p.fixed_std_dev)
        s.scale_normals_to_unity = int(bool(p.scale_normals_to_unity))
        s.device = int(p.device)
        s.dist_rank = int(getattr(p, "dist_rank", 0))
        s.dist_world = int(getattr(p, "dist_world", 1))
        devs = getattr(p, "devices", None)
        if devs:
            self._devs = (C.c_int * len(devs))(*[int(d) for d in devs])
            s.n_devices, s.devices = len(devs), self._devs
        self._transport = _b(getattr(p, "dist_transport", None))
        s.dist_transport = self._transport
        s.dist_two_level = int(bool(getattr(p, "dist_two_level", True)))
        s.defer_variances = int(getattr(p, "defer_variances", 2))
        s.batch_blocks = int(getattr(p, "batch_blocks", 32))
        s.reuse_factors = int(bool(getattr(p, "reuse_factors", True)))
        s.chain_runs = int(getattr(p, "chain_runs", -1))
        return s

    # ---- multi-GPU (include/dnaadjust_c.h "multi-GPU") ----
    def attach_rccl(self, rank, world, unique_id, device):
        self._chk(self.lib.dnaadj_dist_attach_rccl(self.h, int(rank), int(world), bytes(unique_id), int(device)))

    def AdjustNetworkDistributed(self):
        st = C.c_int()
        self._chk(self.lib.dnaadj_adjust_distributed(self.h, C.byref(st)))
        return st.value

    def dist_info(self):
        r, w = C.c_int(), C.c_int()
        buf = C.create_string_buffer(32)
        self.lib.dnaadj_dist_info(self.h, C.byref(r), C.byref(w), buf, 32)
        return r.value, w.value, buf.value.decode()

    def block_owner(self, k):
        return self.lib.dnaadj_block_owner(self.h, k)

    def inverse_exchange_stats(self):
        """intra-block distributed inverse (simultaneous adjustment on several GPUs): split launches, bytes received (this rank)"""
        n, b = C.c_uint64(), C.c_double()
        self.lib.dnagpu_inverse_exchange_stats(self.device_context(), C.byref(n), C.byref(b))
        return {"split_launches": n.value, "bytes_received": b.value}

    def exchange_stats(self):
        b, e, c = C.c_uint64(), C.c_double(), C.c_double()
        self.lib.dnaadj_exchange_stats(self.h, C.byref(b), C.byref(e), C.byref(c))
        return {"bytes": b.value, "exchange_ms": e.value, "chain_ms": c.value}

    def device_instance_context(self, r):
        """a.devices: the device context of GPU r's instance (dnagpu_profile_*)"""
        return self.lib.dnaadj_device_instance_context(self.h, int(r))

    def device_instance_stats(self, r):
        """a.devices: what GPU r's instance did since the last ResetAdjustment (dnaadj_instance_stats as a dict)"""
        st = DnaAdjInstanceStats()
        if self.lib.dnaadj_device_instance_stats(self.h, int(r), C.byref(st)) != 0:
            raise NetAdjustException("dnaadj_device_instance_stats: no such instance")
        return {k: getattr(st, k) for k, _ in DnaAdjInstanceStats._fields_}

    def AdjustNetwork(self):
        st = C.c_int()
        self._chk(self.lib.dnaadj_adjust(self.h, C.byref(st)))
        return st.value

    def ResetAdjustment(self):
        """measurement helper: state right after PrepareAdjustment, data stays resident in HBM"""
        self._chk(self.lib.dnaadj_reset(self.h))

    def CancelAdjustment(self):
        self._chk(self.lib.dnaadj_cancel(self.h))

    def blockCount(self):
        return self.lib.dnaadj_block_count(self.h)

    def CurrentIteration(self):
        return self.lib.dnaadj_iterations(self.h)

    def GetMaxCorrection(self):
        return self.lib.dnaadj_max_correction(self.h)

    def GetIterationCorrection(self, it):
        return self.lib.dnaadj_iteration_correction(self.h, it)

    def GetMeasurementCount(self):
        return self.lib.dnaadj_measurement_count(self.h)

    def GetUnknownsCount(self):
        return self.lib.dnaadj_unknowns_count(self.h)

    def GetDegreesOfFreedom(self):
        return self.lib.dnaadj_degrees_of_freedom(self.h)

    def adjustTime(self):
        """milliseconds spent inside AdjustNetwork()"""
        return self.lib.dnaadj_adjust_time_ms(self.h)

    def GenerateStatistics(self):
        """dna_adjust::GenerateStatistics (dnaadjust.cpp:6802)"""
        self._chk(self.lib.dnaadj_generate_statistics(self.h))

    def _stats(self):
        st = DnaAdjStatistics()
        self._chk(self.lib.dnaadj_get_statistics(self.h, C.byref(st)))
        return st

    def GetChiSquared(self):
        return self._stats().chi_squared

    def GetSigmaZero(self):
        return self._stats().sigma_zero

    def GetGlobalPelzerRel(self):
        return self._stats().global_pelzer

    def GetChiSquaredUpperLimit(self):
        return self._stats().chi_upper_limit

    def GetChiSquaredLowerLimit(self):
        return self._stats().chi_lower_limit

    def GetPotentialOutlierCount(self):
        return self._stats().potential_outliers

    def GetTestResult(self):
        return self._stats().test_result

    def SerialiseAdjustedVarianceMatrices(self):
        """<output_folder>/<network_name>-rva.mtx and -pam.mtx (dnaadjust.cpp:6770)"""
        self._chk(self.lib.dnaadj_serialise_adjusted_variance_matrices(self.h))

    def DeSerialiseAdjustedVarianceMatrices(self):
        self._chk(self.lib.dnaadj_deserialise_adjusted_variance_matrices(self.h))

    def UpdateBinaryFiles(self):
        """rewrites the .bst / .bms with the adjusted values (dnaadjust.cpp:445)"""
        self._chk(self.lib.dnaadj_update_binary_files(self.h))

    # ---- results ---------------------------------------------------------------
    def measurement_records(self):
        """the measurement_t records held by the adjustment, as raw 208-byte records (numpy uint8 [n, 208])"""
        n = self.lib.dnaadj_measurement_record_count(self.h)
        out = np.zeros((n, 208), dtype=np.uint8)
        self._chk(self.lib.dnaadj_measurement_records(self.h, out.ctypes.data_as(C.c_void_p), n))
        return out

    def block_prec_adj_msrs(self, block):
        n = self.lib.dnaadj_block_prec_adj_msrs_count(self.h, block)
        out = np.zeros(n, dtype=np.float64)
        if n:
            self._chk(self.lib.dnaadj_block_prec_adj_msrs(self.h, block, out.ctypes.data_as(c_f64p), n))
        return out

    def block_stations(self, block):
        n = self.lib.dnaadj_block_station_count(self.h, block)
        out = np.empty(n, dtype=np.uint32)
        self._chk(self.lib.dnaadj_block_stations(self.h, block, out.ctypes.data_as(c_u32p)))
        return out

    def block_estimates(self, block):
        n = self.lib.dnaadj_block_station_count(self.h, block)
        out = np.empty(3 * n, dtype=np.float64)
        self._chk(self.lib.dnaadj_block_estimates(self.h, block, out.ctypes.data_as(c_f64p)))
        return out

    def block_variances_packed(self, block):
        n = 3 * self.lib.dnaadj_block_station_count(self.h, block)
        out = np.empty(n * (n + 1) // 2, dtype=np.float64)
        self._chk(self.lib.dnaadj_block_variances_packed(self.h, block, out.ctypes.data_as(c_f64p)))
        return out

    def adjusted_coordinates(self, n_stations):
        out = np.empty(3 * n_stations, dtype=np.float64)
        self._chk(self.lib.dnaadj_adjusted_coordinates(self.h, out.ctypes.data_as(c_f64p)))
        return out.reshape(-1, 3)

    # ---- measurement -------------------------------------------------------------
    def solve_flops(self):
        return self.lib.dnaadj_solve_flops(self.h)

    def solve_count(self):
        return self.lib.dnaadj_solve_count(self.h)

    def algorithmic_flops(self):
        return self.lib.dnaadj_algorithmic_flops(self.h)

    def completion_count(self):
        return self.lib.dnaadj_completion_count(self.h)

    def factor_reuses(self):
        """reuse_factors: block steps of the last AdjustNetwork served from a factor kept since an earlier iteration"""
        return int(self.lib.dnaadj_factor_reuses(self.h))

    def minimal_work_flops(self):
        """of algorithmic_flops(): every factorisation once + the variance matrices once -- nothing that was done again"""
        return float(self.lib.dnaadj_minimal_work_flops(self.h))

    def small_batch_steps(self):
        """of factor_reuses(): block steps that went out as one launch over many small blocks (dnagpu_small_batch_*)"""
        return int(self.lib.dnaadj_small_batch_steps(self.h))

    def chain_runs(self):
        """a.chain_runs: the runs the junction chains are cut into (their steps advance together in merged launches); 0 = step by step"""
        return int(self.lib.dnaadj_chain_runs(self.h))

    def chain_step_reuses(self):
        return int(self.lib.dnaadj_chain_step_reuses(self.h))

    def batched_block_steps(self):
        """block steps that went through batched calls (settings.batch_blocks)"""
        return int(self.lib.dnaadj_batched_block_steps(self.h))

    def oscillation_history(self):
        """UpdateIterationDiagnostics' records: dicts keyed like the reference's OscillationRecord (dnaadjust.hpp:1277-1285)"""
        n = self.lib.dnaadj_oscillation_history(self.h, None, 0)
        out = (C.c_double * (9 * max(1, n)))()
        self.lib.dnaadj_oscillation_history(self.h, out, n)
        keys = ("station", "first_iteration", "last_iteration", "cycles", "first_mag", "last_mag", "last_e", "last_n", "last_up")
        return [{k: (int(out[9 * i + j]) if j < 4 else float(out[9 * i + j])) for j, k in enumerate(keys)} for i in range(n)]

    def summaries(self, limit=20):
        """the text of PrintOscillationSummary() + PrintSuspectMeasurementSummary(limit)"""
        n = self.lib.dnaadj_summaries(self.h, limit, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.lib.dnaadj_summaries(self.h, limit, buf, n + 1)
        return buf.value.decode("utf-8", errors="replace")

    def batched_flops(self):
        return float(self.lib.dnaadj_batched_flops(self.h))

    def memory_plan(self):
        """PrepareAdjustment's memory plan (dnaadj_memory_plan): where the staged variance matrices go, which blocks keep their factor"""
        out = (C.c_double * 12)()
        if self.lib.dnaadj_memory_plan(self.h, out) != 0:
            return {}
        return {"staged_variances_host_bytes": int(out[0]), "staged_variances_packed_in_hbm_bytes": int(out[1]),
                "staged_variances_host_gb": round(out[0] / 1e9, 2), "staged_variances_packed_in_hbm_gb": round(out[1] / 1e9, 2),
                "blocks_keeping_their_factor": int(out[2]), "blocks_condensed": int(out[3]), "batch_members_beyond_first": int(out[4]),
                "host_memory_available_gb": round(out[5] / 1e9, 1),
                "copied_to_host_gb": round(out[6] / 1e9, 2), "waited_for_copies_ms": round(out[7], 1),
                "factors_made_again": int(out[8]), "blocks_without_kept_factor_refactor": bool(out[9]),
                "factors_taken_from_their_packed_copy": int(out[10]), "blocks_packing_their_factor": int(out[11])}

    def condensed_schedule(self):
        """True when the prepared adjustment runs the condensed schedule (phased, schur_carry, a segmentation that fits it)"""
        return bool(self.lib.dnaadj_condensed_schedule(self.h))

    def elimination_count(self):
        return self.lib.dnaadj_elimination_count(self.h)

    def device_context(self):
        return self.lib.dnaadj_device_context(self.h)


def import_dna_text(stn_file, msr_file, out_base, geo_file=None):
    """DNA text station / measurement files (+ optional DNA geoid file) -> out_base.bst / .bms / .asl (GNSS measurements aligned to the
    stations' frame)"""
    from ._lib import DnaImportSummary
    lib = _lib.load()
    out = DnaImportSummary()
    err = C.create_string_buffer(512)
    if lib.dnaimport_text_geo(os.fsencode(stn_file), os.fsencode(msr_file), os.fsencode(geo_file) if geo_file else None, os.fsencode(out_base),
                              C.byref(out), err, 512) != 0:
        raise RuntimeError("dnaimport_text: " + err.value.decode(errors="replace"))
    return {k: getattr(out, k) for k, _ in DnaImportSummary._fields_}


def rccl_unique_id():
    """128 bytes for dnaadj_dist_attach_rccl (ncclGetUniqueId): make on one rank, hand to all"""
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    err = C.create_string_buffer(256)
    if lib.dnaadj_dist_unique_id(buf, err, 256) != 0:
        raise RuntimeError("dnaadj_dist_unique_id: " + err.value.decode(errors="replace"))
    return buf.raw


def write_synthetic_network(folder, name, rows, cols, n_baselines=0, n_blocks=1, seed=20260928, initial_sigma=0.05,
                            x_clusters=0, y_cluster=False, y_llh=False, scalars=False, ragged=0.0, rows_lo=0, rows_hi=0):
    """writes <folder>/<name>.{bst,bms,asl,seg,truth}; returns the summary dict.  ragged / rows_lo / rows_hi: uneven
    strips (include/dnaadjust_c.h dnasynth_spec)."""
    lib = _lib.load()
    spec = DnaSynthSpec(rows, cols, n_baselines, n_blocks, seed, initial_sigma, int(x_clusters), int(bool(y_cluster)), int(bool(y_llh)),
                        int(bool(scalars)), int(rows_lo), int(rows_hi), float(ragged))
    out = DnaSynthSummary()
    err = C.create_string_buffer(512)
    rc = lib.dnasynth_write_network(os.fsencode(folder), os.fsencode(name), C.byref(spec), C.byref(out), err, 512)
    if rc != 0:
        raise RuntimeError("dnasynth_write_network: " + err.value.decode(errors="replace"))
    return {k: getattr(out, k) for k, _ in DnaSynthSummary._fields_}
