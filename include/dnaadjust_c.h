/*
 * dnaadjust_c.h -- C-ABI over the C++ drop-in class dynadjust::networkadjust::dna_adjust
 * (dynadjust_amd/csrc/host/dna_adjust.hpp), which mirrors the reference's
 * dna_adjust entry points (dynadjust/dynadjust/dnaadjust/dnaadjust.hpp:259-405 of
 * /root/reference): PrepareAdjustment / AdjustNetwork / getters.  This is what a non-C++
 * host (the Python tests and bench, or a cgo/JNI caller) binds; the reference's own
 * dnaadjustwrapper would link the C++ class directly.
 *
 * All functions return 0 on success or a negative code; the text of the C++ exception
 * that the reference would have thrown is available through dnaadj_last_error().
 */
#ifndef DNAADJUST_C_H_
#define DNAADJUST_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dnaadj_handle dnaadj_handle;

/* subset of project_settings (include/config/dnaoptions.hpp:425-493) read by the path */
typedef struct {
    const char* bst_file;      /* a.bst_file */
    const char* bms_file;      /* a.bms_file */
    const char* asl_file;      /* s.asl_file (may be NULL: every station valid) */
    const char* seg_file;      /* a.seg_file (phased mode) */
    int adjust_mode;           /* 0 SimultaneousMode, 1 PhasedMode */
    int multi_thread;          /* --multi-thread: forward and reverse chains on separate streams */
    int max_iterations;        /* default 10 */
    float iteration_threshold; /* default 0.0005 */
    double free_std_dev;       /* default 10 */
    double fixed_std_dev;      /* default 1e-6 */
    int scale_normals_to_unity;
    int device;                /* HIP device ordinal of this process */
    float confidence_interval; /* a.confidence_interval, default 95 (global test + outlier flag) */
    int output_tstat;          /* o._adj_msr_tstat: fill measurement_t::TStat in dnaadj_generate_statistics */
    const char* network_name;  /* g.network_name   -> <output_folder>/<network_name>-rva.mtx / -pam.mtx (may be NULL) */
    const char* output_folder; /* g.output_folder (may be NULL = ".") */
    int reuse_inverses;        /* device path only (default 0): phased GNSS-only networks keep the block inverses of the first
                                  iteration in HBM and reuse them afterwards (identical results, half the solves) */
    int schur_carry;           /* device path only (default 1): forward / reverse steps whose solution is only carried to the next
                                  block eliminate the inner unknowns instead of inverting the block (dnagpu_schur_carry; same
                                  carried weights and estimates up to rounding, ~0.35 n^3 instead of n^3 flops).  0 = every step
                                  inverts its block like the reference's Solve().  Ignored with reuse_inverses or
                                  scale_normals_to_unity */
    int stage;                 /* a.stage (--staged-adjustment): the rigorous variance matrices are kept in page-locked host memory
                                  instead of HBM (packed lower triangles); switches itself on when they would not fit on the device */
    int keep_factors;          /* device path only (default 1; needs schur_carry): the condensing step keeps its factor in HBM (two
                                  n x n matrices per block, as far as memory allows) and the rigorous solve of the block completes
                                  it instead of inverting the block's normals again: ~1.0 instead of ~1.36 inverse-equivalents
                                  per block and iteration, same results up to rounding */
    /* ---- multi-GPU (device path only; all zero / NULL = one GPU).  See "multi-GPU" below. ---- */
    int dist_rank;             /* one process per GPU: this process's rank ... */
    int dist_world;            /* ... of dist_world (0 or 1 = single process) */
    int n_devices;             /* one process for several GPUs: how many entries `devices` has (0 or 1 = `device` only) */
    const int* devices;        /* their HIP ordinals */
    const char* dist_transport;/* NULL = choose; "rccl"; "local" (ranks of one process sharing a GPU); "shared" (one PROCESS per rank without
                                  RCCL between them -- several on one GPU, or no fabric: host-staged over TCP, dist_comm_shared.cpp) */
    int dist_two_level;        /* condensed chains across ranks (default 1): 1 = two-level where the blocks are one contiguous network and every
                                  rank owns a run (see dnatypes.hpp; cfg4-sized junction rows: 0.29 s instead of 1.75 s per iteration), 0 = on every rank */
    int defer_variances;       /* condensed schedule with kept factors (default 2): the corrections of an iteration come from every block's
                                  completed factor and the inverses (rigorous variance matrices) are formed once, after the last iteration;
                                  2 = the condensing step also stops at the factor (no inverse of the eliminated part until the end),
                                  0 = an inverse per block and iteration like dna_adjust::Solve */
    int batch_blocks;          /* condensed schedule with kept factors (default 32 = DNAGPU_BATCH_MAX): blocks of one shape go through its large
                                  steps as one batch of up to this many members -- merged launches, in lock step (include/dnagpu.h,
                                  dnagpu_*_batched); every member's results are the bits of the unbatched calls.  0 / 1 = off */
    int reuse_factors;         /* device path only (default 1): GNSS-only networks (the reference's own test, dnaadjust.cpp:2457, which it
                                  applies in simultaneous mode) keep the factors of iteration 1 -- the blocks' light factors, the kept
                                  blocks' factors, the chain steps' factors on the condensed blocks -- and renew right-hand sides only
                                  from iteration 2 on; the variance matrices are formed once, after the last iteration, as before.
                                  0 = every iteration factors again.  Needs schur_carry, keep_factors and defer_variances = 2 */
    int chain_runs;            /* condensed schedule on ONE GPU, many small blocks (a dnasegment-default cut; default -1 = choose): the two
                                  junction chains are cut into this many runs whose steps advance together in merged launches
                                  (dna_adjust::LockstepChains, dnagpu_chain_plan_*): 2 B / W + W steps deep instead of B.  0 / 1 = one
                                  run (the chains step by step, block after block) */
} dnaadj_settings;

#define DNAADJ_OK 0
#define DNAADJ_EXCEPTION (-1)   /* the reference would have thrown; see dnaadj_last_error() */
#define DNAADJ_EINVAL (-2)

void dnaadj_default_settings(dnaadj_settings* s);
int dnaadj_create(dnaadj_handle** out);
void dnaadj_destroy(dnaadj_handle* h);
const char* dnaadj_last_error(const dnaadj_handle* h);

int dnaadj_prepare(dnaadj_handle* h, const dnaadj_settings* s);        /* dna_adjust::PrepareAdjustment */
/* PrepareAdjustment's plan for `world` GPUs of `hbm_bytes` each as JSON, WITHOUT a device (a dry run for a node one does not have, and the
 * device-free view of the N > 1 schedule): per rank the owned run of blocks and its share of sum n^3, the HBM budget (blocks and chain
 * data, chains' workspaces, variance matrices or the staged store's host / device split, kept factors, batch workspaces) and whether it
 * fits, the two-level chains' run (end stations, merges in order, steps per level) and the bytes of every exchange of an iteration
 * (condensed blocks one-level, run systems two-level, the coordinates' all-reduce).  The handle is left unprepared.
 * (the reference sizes its threads' work the same way before it starts them: dnaadjust-multi.cpp:92-140) */
int dnaadj_plan_distributed(dnaadj_handle* h, const dnaadj_settings* s, int world, double hbm_bytes, char* json, size_t cap, size_t* needed);
int dnaadj_adjust(dnaadj_handle* h, int* status);                      /* dna_adjust::AdjustNetwork -> _ADJUST_STATUS_ */
int dnaadj_cancel(dnaadj_handle* h);                                   /* dna_adjust::CancelAdjustment */
/* measurement helper (not in the reference): back to the state right after dnaadj_prepare, data stays in HBM */
int dnaadj_reset(dnaadj_handle* h);

uint32_t dnaadj_block_count(const dnaadj_handle* h);                   /* blockCount() */
uint32_t dnaadj_iterations(const dnaadj_handle* h);                    /* CurrentIteration() */
double dnaadj_max_correction(const dnaadj_handle* h);                  /* GetMaxCorrection() */
double dnaadj_iteration_correction(const dnaadj_handle* h, uint32_t iteration);
uint32_t dnaadj_measurement_count(const dnaadj_handle* h);             /* GetMeasurementCount() */
uint32_t dnaadj_unknowns_count(const dnaadj_handle* h);                /* GetUnknownsCount() */
int dnaadj_degrees_of_freedom(const dnaadj_handle* h);                 /* GetDegreesOfFreedom() */
double dnaadj_adjust_time_ms(const dnaadj_handle* h);                  /* adjustTime() */
double dnaadj_solve_flops(const dnaadj_handle* h);                     /* sum n^3 over Solve() calls */
double dnaadj_algorithmic_flops(const dnaadj_handle* h);               /* n^3 per inverse; n_i^3/3 + n_i^2 n_j + n_i n_j^2 + n_j^3 per elimination step */
uint32_t dnaadj_solve_count(const dnaadj_handle* h);
uint32_t dnaadj_completion_count(const dnaadj_handle* h);              /* of those, rigorous solves that completed a kept factor (keep_factors) */
uint32_t dnaadj_elimination_count(const dnaadj_handle* h);             /* of those, carry-only steps done by elimination (schur_carry) */
/* reuse_factors: block steps (condensing + rigorous solve counted once) / chain steps of the last dnaadj_adjust that were served from a
 * factor kept since an earlier iteration -- right-hand sides only (this GPU's share) */
uint64_t dnaadj_factor_reuses(const dnaadj_handle* h);
/* of dnaadj_algorithmic_flops, the flops of the minimal schedule: every factorisation once and the variance matrices once (GNSS-only
 * networks; with terrestrial measurements every iteration's factorisations count) -- nothing that was done again */
double dnaadj_minimal_work_flops(const dnaadj_handle* h);
uint64_t dnaadj_chain_step_reuses(const dnaadj_handle* h);
/* ... of the block steps, those that went out as one launch over many small blocks (dnagpu_small_batch_*: networks of 32 and more blocks
 * of up to ~680 stations each -- a dnasegment-default cut) */
uint64_t dnaadj_small_batch_steps(const dnaadj_handle* h);
/* a.chain_runs: the runs the junction chains of this network are cut into (steps advancing together, dna_adjust::LockstepChains); 0 = step by step */
int dnaadj_chain_runs(const dnaadj_handle* h);
uint32_t dnaadj_station_count(const dnaadj_handle* h);

uint32_t dnaadj_block_station_count(const dnaadj_handle* h, uint32_t block);
int dnaadj_block_stations(dnaadj_handle* h, uint32_t block, uint32_t* stations);           /* global ids, ascending */
int dnaadj_block_estimates(dnaadj_handle* h, uint32_t block, double* xyz);                 /* v_rigorousStations_ */
int dnaadj_block_variances_packed(dnaadj_handle* h, uint32_t block, double* packed);       /* v_rigorousVariances_ */
int dnaadj_adjusted_coordinates(dnaadj_handle* h, double* xyz);                            /* 3 per bst station */

/* ---- after the adjustment: statistics and results out ---------------------------------- */
typedef struct {
    double chi_squared;        /* GetChiSquared()            dnaadjust.hpp:339 */
    double sigma_zero;         /* GetSigmaZero()             dnaadjust.hpp:340 */
    double global_pelzer;      /* GetGlobalPelzerRel()       dnaadjust.hpp:350 */
    double chi_upper_limit;    /* GetChiSquaredUpperLimit()  dnaadjust.hpp:344 */
    double chi_lower_limit;    /* GetChiSquaredLowerLimit()  dnaadjust.hpp:347 */
    uint32_t measurement_params, unknown_params;
    uint32_t potential_outliers;   /* GetPotentialOutlierCount() dnaadjust.hpp:341 */
    uint32_t test_result;          /* GetTestResult(): 0 pass, 1 warning, 2 fail  dnaadjust.hpp:353 */
    int degrees_of_freedom;
} dnaadj_statistics;
int dnaadj_generate_statistics(dnaadj_handle* h);                          /* dna_adjust::GenerateStatistics (dnaadjust.cpp:6802) */
int dnaadj_get_statistics(const dnaadj_handle* h, dnaadj_statistics* out);
/* the measurement records (measurement_t, 208 bytes each, include/measurement_types/dnameasurement.hpp:133-194) as
 * the adjustment holds them: after dnaadj_generate_statistics with measAdj, measCorr, measAdjPrec, residualPrec,
 * NStat, TStat, PelzerRel filled and the variances scaled */
uint64_t dnaadj_measurement_record_count(const dnaadj_handle* h);
int dnaadj_measurement_records(const dnaadj_handle* h, void* records, uint64_t cap_records);
/* v_precAdjMsrsFull_ of a block: 6 doubles (xx xy xz yy yz zz) per GNSS vector in CML order */
uint64_t dnaadj_block_prec_adj_msrs_count(const dnaadj_handle* h, uint32_t block);
int dnaadj_block_prec_adj_msrs(const dnaadj_handle* h, uint32_t block, double* out, uint64_t cap);
int dnaadj_serialise_adjusted_variance_matrices(dnaadj_handle* h);         /* SerialiseAdjustedVarianceMatrices (dnaadjust.cpp:6770) */
int dnaadj_deserialise_adjusted_variance_matrices(dnaadj_handle* h);       /* DeSerialiseAdjustedVarianceMatrices (dnaadjust.cpp:6720) */
int dnaadj_update_binary_files(dnaadj_handle* h);                          /* UpdateBinaryFiles (dnaadjust.cpp:445) */

/* quantiles used by the global test (the reference takes them from boost::math, dnaadjust.cpp:203-206, :6866-6880) */
double dnastat_normal_quantile(double p);
double dnastat_chi_squared_quantile(double dof, double p);

/* ---- per-block steps of the phased chain (dna_adjust::Phased*): what the multi-GPU orchestrator schedules.
 * AdjustPhasedForward / AdjustPhasedReverseCombine (dnaadjust.cpp:2756, 3461) are loops over exactly these. ---- */
int dnaadj_block_flags(const dnaadj_handle* h, uint32_t block, int* first, int* last, int* isolated);   /* blockMeta_t */
uint32_t dnaadj_junction_unknowns(const dnaadj_handle* h, uint32_t block);                              /* 3*|JSL(block)| */
size_t dnaadj_junction_payload_doubles(const dnaadj_handle* h, uint32_t block);
int dnaadj_phased_begin_iteration(dnaadj_handle* h);
int dnaadj_phased_forward_block(dnaadj_handle* h, uint32_t block, double* max_corr);
int dnaadj_phased_reverse_block(dnaadj_handle* h, uint32_t block, double* max_corr);
int dnaadj_phased_combine_block(dnaadj_handle* h, uint32_t block, double* max_corr);
int dnaadj_phased_finalise_block(dnaadj_handle* h, uint32_t block);
int dnaadj_phased_note_correction(dnaadj_handle* h, double max_corr);
int dnaadj_phased_end_iteration(dnaadj_handle* h, int* iterate);       /* convergence test + UpdateAdjustment */
int dnaadj_phased_finish(dnaadj_handle* h, int* status);               /* ValidateandFinaliseAdjustment */
/* kind 0 = forward (v_junctionVariancesFwd_/v_junctionEstimatesFwd_ of `block`), 1 = reverse; buf may be device memory */
/* The condensed schedule (settings.schur_carry; dna_adjust_phased.cpp): per iteration
 *   (A) dnaadj_phased_condense_block(k) for every block, in any order, on any process   -- eliminates the stations of block k
 *       that no other block holds; the result (dnaadj_condensed_export / _import) must reach every process that runs (B)
 *   (B) dnaadj_phased_condensed_forward(k), k ascending, and dnaadj_phased_condensed_reverse(k), k descending: the two
 *       chains of the reference on the condensed blocks; they leave the junction weights / estimates of every block
 *   (C) dnaadj_phased_rigorous_block(k) for every block, in any order, on any process that ran (B): the one full solve per
 *       block whose result is rigorous, finalised and noted for the convergence test
 * dnaadj_condensed_schedule() tells whether the prepared adjustment supports it (phased, schur_carry, no reuse_inverses /
 * scale_normals_to_unity). */
int dnaadj_staged(const dnaadj_handle* h);     /* 1 when the prepared adjustment keeps its rigorous variances in host memory */
/* the memory plan PrepareAdjustment made (DecideStaging / PrepareCondensedBlocks), own blocks only: out[0] bytes of staged variance matrices in
 * page-locked host memory, out[1] bytes of them packed in HBM (past the host's memory limit), out[2] blocks that may keep their factor
 * between the condensing step and the rigorous solve, out[3] blocks that have something to condense, out[4] members beyond the first
 * a batch of blocks may have, out[5] bytes of host memory the process could still take when the plan was made, out[6] bytes the last adjustment copied to host memory
 * (staged variance matrices), out[7] milliseconds it waited for those copies (FinishStagedCopies: what did not hide behind the products), out[8] how many times
 * the last adjustment made a block's factor AGAIN because the block may not keep one (rigorous solves + variance matrices), out[9] 1 when
 * that is the plan for the blocks without a kept factor (GNSS-only network, light factors) instead of an inverse per iteration,
 * out[10] how many times it took such a factor from its packed copy in HBM instead (the slot that waits for the block's packed
 * variance matrix lends itself to the factor during the iterations), out[11] how many blocks do that */
/* Oscillation diagnostics (dna_adjust::UpdateIterationDiagnostics / PrintOscillationSummary / PrintSuspectMeasurementSummary,
 * ADJ:7450-7780).  dnaadj_oscillation_history: the recorded stations, 9 doubles each (bst index, first iteration, last iteration, cycles,
 * first magnitude, last magnitude, last e / n / up), ascending by station; returns their number (cap_records limits what is written).
 * dnaadj_summaries: what the two Print...Summary members write (oscillation summary, then suspect measurements with `limit` lines per
 * list), as text; returns the length needed. */
size_t dnaadj_oscillation_history(const dnaadj_handle* h, double* out9, size_t cap_records);
size_t dnaadj_summaries(dnaadj_handle* h, size_t limit, char* buf, size_t cap);
int dnaadj_memory_plan(const dnaadj_handle* h, double out[12]);
int dnaadj_condensed_schedule(const dnaadj_handle* h);
/* Across GPUs every wait for the other ranks has a deadline (default 600 s, DNAGPU_COLLECTIVE_TIMEOUT_S): past it the communicator is
 * aborted (ncclCommAbort) and AdjustNetwork() ends with ADJUST_EXCEPTION_RAISED on the ranks that are still alive -- the reference's
 * threads unblock each other with a sentinel (dnaadjust-multi.cpp:36-58, 182-190).  Process-wide. */
void dnaadj_dist_set_timeout(double seconds);
/* test hook: rank `rank` sleeps `seconds` before its n-th agreement with the others (a rank hanging in a kernel, seen from outside) */
void dnaadj_debug_stall_rank(int rank, long nth_agreement, double seconds);
/* how many block steps of the last adjustment went through batched calls (settings.batch_blocks): condensing and rigorous solve count a
 * block once per iteration, the variance matrices once */
uint64_t dnaadj_batched_block_steps(const dnaadj_handle* h);
double dnaadj_batched_flops(const dnaadj_handle* h);       /* the algorithmic flops of those steps (of dnaadj_algorithmic_flops) */
size_t dnaadj_condensed_payload_doubles(const dnaadj_handle* h, uint32_t block);
int dnaadj_phased_condense_block(dnaadj_handle* h, uint32_t block);
int dnaadj_phased_condensed_forward(dnaadj_handle* h, uint32_t block);
int dnaadj_phased_condensed_reverse(dnaadj_handle* h, uint32_t block);
int dnaadj_phased_rigorous_block(dnaadj_handle* h, uint32_t block, double* max_corr);
/* the same for lists of blocks / both chains at once: with settings.multi_thread the work is spread over the two device chains */
int dnaadj_phased_condense_blocks(dnaadj_handle* h, const uint32_t* blocks, size_t n);
int dnaadj_phased_condensed_chains(dnaadj_handle* h);
int dnaadj_phased_rigorous_blocks(dnaadj_handle* h, const uint32_t* blocks, size_t n);
int dnaadj_condensed_export(dnaadj_handle* h, uint32_t block, double* buf);
int dnaadj_condensed_import(dnaadj_handle* h, uint32_t block, const double* buf);
/* GenerateStatistics across processes (each holds the rigorous variances of its own blocks): _prepare on every process;
 * _blocks(own blocks); sum the partial chi-square / outlier counts and the per-record arrays (9 doubles per .bms record:
 * touched, measAdj, measCorr, measAdjPrec, residualPrec, NStat, PelzerRel, preAdjCorr, term1; zero where not computed here) over
 * the processes and hand the sums back (_set_partial, _record_statistics_set); _finish on every process.  Afterwards the
 * getters of dnaadj_get_statistics / dnaadj_measurement_records answer as after dnaadj_generate_statistics on one process. */
int dnaadj_statistics_prepare(dnaadj_handle* h);
int dnaadj_statistics_blocks(dnaadj_handle* h, const uint32_t* blocks, size_t n);
int dnaadj_statistics_get_partial(const dnaadj_handle* h, double* chi_squared, uint32_t* outliers);
int dnaadj_statistics_set_partial(dnaadj_handle* h, double chi_squared, uint32_t outliers);
int dnaadj_record_statistics_get(const dnaadj_handle* h, double* out9, uint64_t cap_records);
int dnaadj_record_statistics_set(dnaadj_handle* h, const double* in9, uint64_t n_records);
int dnaadj_statistics_finish(dnaadj_handle* h);
int dnaadj_junction_export(dnaadj_handle* h, int kind, uint32_t block, double* buf);
int dnaadj_junction_import(dnaadj_handle* h, int kind, uint32_t block, const double* buf);
/* which: 0 original, 1 estimated, 2 rigorous */
int dnaadj_block_get_coords(dnaadj_handle* h, uint32_t block, int which, double* xyz);
int dnaadj_block_set_coords(dnaadj_handle* h, uint32_t block, const double* xyz);   /* original = estimated = rigorous */
int dnaadj_block_recompute_b(dnaadj_handle* h, uint32_t block);

/* ---- multi-GPU: the reference's parallel driver is a member of the class (dna_adjust::AdjustPhasedMultiThread,
 * dnaadjust-multi.cpp:92-244); here the blocks of one network are spread over GPUs and the driver stays inside AdjustNetwork().
 *   one process per GPU   settings.dist_rank / dist_world (+ settings.device) on every process.  The RCCL communicator is made
 *                         inside dnaadj_prepare (rank 0's unique id reaches the others over TCP, MASTER_ADDR : MASTER_PORT + 17 or
 *                         DNAGPU_MASTER_PORT), or beforehand by the host: dnaadj_dist_unique_id on one rank, the 128 bytes to
 *                         every rank by whatever the host has (MPI_Bcast, a torch.distributed broadcast, a file), then
 *                         dnaadj_dist_attach_rccl on every rank (collective: ncclCommInitRank).
 *   one process, N GPUs   settings.devices / n_devices: one host thread per GPU inside the library.
 * dnaadj_adjust / dnaadj_adjust_distributed, dnaadj_generate_statistics, dnaadj_serialise_adjusted_variance_matrices,
 * dnaadj_update_binary_files and dnaadj_reset are then collective: call them on every process; rank 0 writes the files. */
#define DNAADJ_UNIQUE_ID_BYTES 128
int dnaadj_dist_rccl_available(void);
int dnaadj_dist_unique_id(unsigned char* id128, char* err, size_t errlen);
int dnaadj_dist_attach_rccl(dnaadj_handle* h, int rank, int world, const unsigned char* id128, int device);
/* AdjustNetwork() of an adjustment prepared across GPUs; DNAADJ_EINVAL if it was not (dnaadj_adjust works for both) */
int dnaadj_adjust_distributed(dnaadj_handle* h, int* status);
/* rank, world, transport ("rccl" / "local" / "none") of the prepared adjustment; rank whose GPU holds a block */
int dnaadj_dist_info(const dnaadj_handle* h, int* rank, int* world, char* transport, size_t len);
int dnaadj_block_owner(const dnaadj_handle* h, uint32_t block);
/* since the last dnaadj_reset: payload bytes this rank moved through the transport, host time in the exchange steps and in
 * the chains on the condensed blocks (ms) */
int dnaadj_exchange_stats(const dnaadj_handle* h, uint64_t* bytes, double* exchange_ms, double* chain_ms);
/* settings.devices: device context of GPU r's instance (r = 0 .. n_devices-1), for dnagpu_profile_* */
void* dnaadj_device_instance_context(dnaadj_handle* h, int r);

/* settings.devices: what GPU r's instance did since the last dnaadj_reset (the getters above answer for the whole process) */
typedef struct {
    int rank, device;                /* its rank in the communicator, its HIP ordinal */
    int rccl_ranks;                  /* ncclCommCount of its communicator (0: transport "local" / none) */
    uint32_t solves, eliminations, completions;
    double algorithmic_flops, solve_flops;
    uint64_t exchanged_bytes;
    double exchange_ms, chain_ms;
} dnaadj_instance_stats;
int dnaadj_device_instance_stats(dnaadj_handle* h, int r, dnaadj_instance_stats* out);
/* test hook: CancelAdjustment() as ONE rank of a multi-process adjustment would receive it (a signal to its process): only GPU r's
 * instance is told; the ranks agree on the cancellation at the next phase boundary and all return ADJUST_CANCELLED */
int dnaadj_debug_cancel_instance(dnaadj_handle* h, int r);
/* test hook: the TCP hand-off of the 128 bytes by itself (what dnaadj_prepare does between processes before ncclCommInitRank): rank 0
 * listens on addr : port and serves ranks 1 .. world - 1 once each, the others connect and read */
int dnaadj_debug_tcp_share_unique_id(int rank, int world, unsigned char* id128, const char* addr, int port, double timeout_s, char* err, size_t errlen);

/* device context of the adjustment (for dnagpu_profile_*), NULL before prepare */
void* dnaadj_device_context(dnaadj_handle* h);

/* ---- DNA text files in: <stn>, <msr> (DNA v3 fixed-column text; stations LLH / LLh / XYZ, GNSS measurements G / X / Y)
 * -> <out_base>.bst / .bms / .asl in dnaimport's record layout.  GNSS measurements given in another frame / epoch than the
 * stations (ITRF1997 ... ITRF2020 -> GDA2020) are aligned the way dnareftran does it: both ends of a baseline transformed as
 * points with the published 14-parameter set at the measurement's epoch, then differenced (dnareftran.cpp:1740-1835). ---- */
typedef struct {
    uint64_t stations, records, vectors, clusters, vectors_transformed;
} dnaimport_summary;
int dnaimport_text(const char* stn_file, const char* msr_file, const char* out_base, dnaimport_summary* out, char* err, size_t errlen);
/* the same with a DNA geoid file (station, N, deflections: what dnageoid exports; may be NULL) and the terrestrial measurement types
 * A B C E H K L M R S V Z, UTM stations, LLH point clusters: the reference's urban sample (sampleData/urban-network.{stn,msr,geo}) */
int dnaimport_text_geo(const char* stn_file, const char* msr_file, const char* geo_file, const char* out_base, dnaimport_summary* out, char* err,
                       size_t errlen);

/* ---- synthetic networks: writes <dir>/<name>.{bst,bms,asl,seg,truth} ---- */
typedef struct {
    uint32_t rows, cols;
    uint64_t n_baselines;   /* 0 = all E/N/NE neighbours */
    uint32_t n_blocks;
    uint64_t seed;
    double initial_sigma;
    uint32_t x_clusters;    /* baselines leaving each of the first x_clusters stations become one 'X' cluster */
    uint32_t y_cluster;     /* 1: datum from 'Y' point clusters over the corner stations (which become FFF) */
    uint32_t y_llh;         /* 1: those clusters in latitude / longitude / height ("LLh", "LLH") with geographic variances */
    uint32_t scalars;       /* 1: variance scalars (v, phi, lambda, h) on part of the measurements */
    /* uneven segmentations (what dnasegment makes of a real network, dnasegment.cpp:235-348; default threshold 150 stations per block,
     * dnaoptions.hpp:381-382): ragged in [0, 1): strip heights proportional to 1 + ragged * U(-1, 1); rows_hi > 0: heights drawn from
     * [rows_lo, rows_hi] rows until the grid is used up -- n_blocks is then a result (dnasynth_summary.blocks) */
    uint32_t rows_lo, rows_hi;
    double ragged;
} dnasynth_spec;
typedef struct {
    uint64_t stations, baselines, measurement_rows, blocks, max_block_unknowns;
} dnasynth_summary;
int dnasynth_write_network(const char* dir, const char* name, const dnasynth_spec* spec, dnasynth_summary* out, char* err,
                           size_t errlen);

/* ---- file format helpers for tests (byte-exact readers of the reference's formats) ---- */
/* returns record counts; any pointer may be NULL */
int dnaio_file_summary(const char* bst, const char* bms, const char* asl, uint64_t* n_stn, uint64_t* n_msr, uint64_t* n_asl, char* err,
                       size_t errlen);
/* reads a .seg with the product's reader; per block 8 values: network id, junction count, inner count,
 * measurement count, design rows, first inner, first junction, first measurement (UINT32_MAX if none) */
int dnaio_seg_summary(const char* seg_path, const char* bms_path, uint32_t* n_blocks, uint32_t* per_block8, uint32_t cap_blocks,
                      char* err, size_t errlen);
size_t dnaio_sizeof_station(void);
size_t dnaio_sizeof_measurement(void);

#ifdef __cplusplus
}
#endif
#endif
