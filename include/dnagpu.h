/*
 * dnagpu.h -- C-ABI of the MI355X (gfx950) device layer behind DynAdjust's
 * dna_adjust hot path.  Plain pointers and sizes only; no exceptions cross this
 * boundary (every call returns 0 or a negative DNAGPU_E* code and leaves a
 * message in dnagpu_last_error()).
 *
 * What each entry point replaces in the reference (paths under
 * /root/reference/dynadjust/):
 *
 *   L2 seam (math::matrix_2d, include/math/dnamatrix_contiguous.{hpp,cpp})
 *     dnagpu_cholesky_inverse_packed  matrix_2d::cholesky_inverse, packed path   dnamatrix_contiguous.cpp:952-991
 *                                     (+ Solve()'s scale_normals_to_unity        dnaadjust/dnaadjust.cpp:6614-6645)
 *     dnagpu_multiply_sym_packed      matrix_2d::multiply_sym (cblas_dspmv)      dnamatrix_contiguous.cpp:1471-1510
 *
 *   Block layer (dna_adjust, dynadjust/dnaadjust/dnaadjust.cpp)
 *     dnagpu_block_create / _set_stations / _set_baselines
 *                                     PrepareStationandVarianceMatrices :701, PrepareDesignAndMsrMnsCmpMatrices :798,
 *                                     LoadVarianceMatrix_G :4214 (W = V^-1 per baseline)
 *     dnagpu_block_compute_b          UpdateDesignMeasMatrices_GX :5283 / AddMsrtoMeasMinusComp :4719
 *     dnagpu_form_normals             UpdateNormals :1364 / UpdateNormals_G :1664 (3x3 scatter into packed N)
 *     dnagpu_add_diag3x3              AddConstraintStationstoNormals{Forward,Reverse,Combine,Simultaneous} :1884-2037
 *     dnagpu_form_rhs                 Solve(): At_Vinv_m = AtVinv * measMinusComp (dense dgemm in the reference) :6659
 *     dnagpu_invert                   Solve(): FormInverseVarianceMatrix :6628 / :8472
 *     dnagpu_solve_corrections        Solve(): corrections = N^-1 * At_Vinv_m :6663-6667
 *     dnagpu_update_estimates         UpdateEstimates{Forward,Reverse,Combine} :3022/:3678/:3718, compute_maximum_value
 *     dnagpu_junction_gather/_invert/_scatter/_rhs
 *                                     CarryStnEstimatesandVariances{Forward,Reverse,Combine} :998/:1133/:3196
 *     dnagpu_download_* / dnagpu_save_rigorous
 *                                     UpdateEstimatesFinal :3744 (v_rigorousStations_, v_rigorousVariances_)
 *
 * Symmetric matrices cross the boundary in the reference's packed-lower
 * column-major layout: element (i >= j) at j*n - j*(j-1)/2 + (i-j)
 * (matrix_2d::packed_index, dnamatrix_contiguous.hpp:363).
 *
 * Threading: a ctx is bound to one device; calls on one ctx must come from one
 * host thread at a time.  Work is stream-ordered per "chain" (0 = forward,
 * 1 = reverse/combine) so the facade can run both chains concurrently.
 */
#ifndef DNAGPU_H_
#define DNAGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dnagpu_ctx dnagpu_ctx;

#define DNAGPU_OK 0
#define DNAGPU_EINVAL (-1)
#define DNAGPU_ENOMEM (-2)
#define DNAGPU_EHIP (-3)
#define DNAGPU_ENOTPOSDEF (-4) /* dpotrf-style failure; column in dnagpu_last_info() */
#define DNAGPU_ENODEVICE (-5)
#define DNAGPU_ETOOLARGE (-6) /* a small-system entry point given a system beyond its limits: take the general calls (no error text) */

#define DNAGPU_NUM_CHAINS 8        /* chains a context provides */
#define DNAGPU_DEFAULT_CHAINS 4    /* chains the facade uses by itself (DNAGPU_CHAINS = 2 .. DNAGPU_NUM_CHAINS overrides) */

/* ---- context ------------------------------------------------------------ */
int dnagpu_create(int device, dnagpu_ctx** out);
void dnagpu_destroy(dnagpu_ctx* ctx);
const char* dnagpu_last_error(const dnagpu_ctx* ctx);
int dnagpu_last_info(const dnagpu_ctx* ctx);
int dnagpu_device_count(void);
/* page-locked host memory (full-rate, asynchronous transfers): the spill area of matrices that do not fit in HBM */
int dnagpu_host_alloc(dnagpu_ctx* ctx, size_t bytes, void** out);
void dnagpu_host_free(dnagpu_ctx* ctx, void* p);
/* plain device buffers for the host's own exchange step (the coordinate vector of the all-reduce, the statistics' partial sums) and
 * synchronous copies between any two of host / device memory */
int dnagpu_device_alloc(dnagpu_ctx* ctx, size_t bytes, void** out);
void dnagpu_device_free(dnagpu_ctx* ctx, void* p);
int dnagpu_copy(dnagpu_ctx* ctx, void* dst, const void* src, size_t bytes);
/* free / total device memory in bytes (hipMemGetInfo) */
int dnagpu_mem_info(dnagpu_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* wait for every stream of the ctx */
int dnagpu_sync(dnagpu_ctx* ctx);

/* ---- L2 seam: matrix_2d on host packed storage --------------------------- */
/* ap (n(n+1)/2 doubles, packed lower) is replaced by its inverse. */
int dnagpu_cholesky_inverse_packed(dnagpu_ctx* ctx, double* ap, uint32_t n, int scale_to_unity);
/* y = A x for packed-lower symmetric A. */
int dnagpu_multiply_sym_packed(dnagpu_ctx* ctx, const double* ap, const double* x, double* y, uint32_t n);

/* ---- profiling ------------------------------------------------------------ */
/* enable/disable HIP-event timing of the tile-GEMM kernel: one event pair on the launch stream around every run of
 * consecutive GEMM launches */
int dnagpu_profile_enable(dnagpu_ctx* ctx, int on);
int dnagpu_profile_reset(dnagpu_ctx* ctx);
/* HBM-side roofline of the large bandwidth-bound kernels: with the switch on, every launch of one of the kinds below that moves at
 * least 1 MB is bracketed by HIP events on its chain's stream.  dnagpu_profile_hbm_get drains the chains and returns, per kind, the
 * ALGORITHMIC bytes (what the launch has to move: matrix read / written once), the summed event durations and the launch count since
 * the last reset.  A bracketed duration is the kernel's own only while nothing else runs on the device: measure with one chain. */
enum {
    DNAGPU_HBM_UNPERMUTE = 0,     /* the completed inverse back to the block's unknown order: np^2 read + np^2 written */
    DNAGPU_HBM_FORM_ORDERED = 1,  /* normals formed in the elimination's order: the lower triangle written once (zero fill + 3x3 blocks) */
    DNAGPU_HBM_SUBSTITUTION = 2,  /* corrections from a kept factor, all launches of one solve: the factor (np^2 / 2) read twice */
    DNAGPU_HBM_SYMV = 3,          /* corrections = N^-1 rhs (multiply_sym MAT:1471): n x np read once */
    DNAGPU_HBM_PACK = 4           /* packed lower triangle out of a square matrix: n^2 / 2 read + n^2 / 2 written */
};
int dnagpu_profile_hbm_enable(dnagpu_ctx* ctx, int on);
int dnagpu_profile_hbm_get(dnagpu_ctx* ctx, double bytes[8], double ms[8], uint64_t count[8], int reset);

/* gemm_flops: flops actually issued; gemm_ms: time during which at least one timed GEMM run was executing (union over
 * the chains' streams: equals the summed run durations with one chain); launches */
int dnagpu_profile_get(dnagpu_ctx* ctx, double* gemm_flops, double* gemm_ms, uint64_t* launches);

/* ---- diagnostics ------------------------------------------------------------ */
/* Error-path testing: the nth (1-based) internal table allocation from now on fails as if the device were out of memory
 * (0 = off).  An inverse / elimination that hits it returns DNAGPU_ENOMEM -- never DNAGPU_OK with a skipped launch. */
int dnagpu_debug_fail_allocation(long nth);
/* tests: the next n allocations of a batch's member workspaces (dnagpu_batch_reserve, dnagpu_*_batched) fail with DNAGPU_ENOMEM as if HBM
 * were full -- the caller must then run the blocks one at a time, with the same results */
int dnagpu_debug_fail_batch_workspaces(long n);
/* Launches with fewer than `tiles` 128 x 128 tiles use the 64-tile latency kernel (default 512); 0 sends every launch through the
 * 128-tile throughput kernel (gemm_f64_dma_kernel), a negative value restores the default.  Returns the previous value. */
long dnagpu_debug_set_small_tiles(long tiles);
/* ... and below `tiles` on 32 x 32 tiles (default 64; 0: never; < 0: the default again): the products of the recursion's bottom and of the chains on
 * condensed blocks, where a launch has too few 64-tiles to occupy the chip.  Same bits.  Returns the previous value. */
long dnagpu_debug_set_tiny_tiles(long tiles);
/* dnagpu_schur_carry's result: 1 (default) information form, 0 estimates form (see there); returns the previous value.  The setting is taken
 * over by the contexts created AFTER the call and stays with a context for its lifetime (dnagpu_info_carry(ctx); NULL: the value a new
 * context would get): contexts running side by side never see each other's form. */
int dnagpu_debug_set_info_carry(int on);
int dnagpu_info_carry(const dnagpu_ctx* ctx);
/* the same switch on one context (between adjustments: junction matrices made in one form are not read in the other); returns the previous value */
int dnagpu_ctx_set_info_carry(dnagpu_ctx* ctx, int on);
/* The workgroup -> tile table a launch of this shape would use (host computation, no device needed): `out` receives up to `cap`
 * entries (it << 16 | jt, 0xffffffff = idle), *per_workgroup = entries per workgroup (1);
 * jt_lo / jt_hi = -1 or the column range of one rank of a split launch.  Returns the number of entries the table has. */
long dnagpu_debug_tile_order(int mt, int nt, int K, int kmode, int lower, int tile, int jt_lo, int jt_hi, uint32_t* out, long cap, int* per_workgroup);
/* ---- device-resident work matrices ----------------------------------------
 * A work matrix is an np x np (np = ceil(n/128)*128) column-major buffer that
 * holds N, then N^-1.  Each chain owns one; junction matrices get their own. */
typedef struct dnagpu_matrix dnagpu_matrix;
typedef struct dnagpu_partial dnagpu_partial;   /* a block between dnagpu_block_reduce(keep) and dnagpu_partial_complete */
int dnagpu_matrix_create(dnagpu_ctx* ctx, uint32_t n_max, dnagpu_matrix** out);
void dnagpu_matrix_destroy(dnagpu_ctx* ctx, dnagpu_matrix* m);
/* logical order n (<= n_max) + zero fill + identity padding */
int dnagpu_matrix_reset(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, uint32_t n);
int dnagpu_matrix_upload_packed(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* ap, uint32_t n);
int dnagpu_matrix_download_packed(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* ap);
/* The same, without holding the chain: the triangle is packed into a per-chain device staging buffer on the chain's stream and
 * leaves on a copy stream of its own; the chain's next kernels overlap the transfer.  `ap` must be page-locked (dnagpu_host_alloc)
 * and is complete after dnagpu_copies_sync() or dnagpu_sync().  Falls back to the synchronous call when the staging buffer does
 * not fit.  (The reference's --staged-adjustment writes each block's matrices to its memory-mapped file inside
 * SerialiseBlockToMappedFile, dnaadjust-stage.cpp: on the path of the next block.) */
int dnagpu_matrix_download_packed_async(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* ap);
int dnagpu_copies_sync(dnagpu_ctx* ctx);
/* The staged store when host memory is the scarce side (a container whose memory limit is below the network's variance matrices, e.g.
 * cfg4 on one GPU: 373 GB of packed triangles against a 300 GiB cgroup): the same packed lower triangle, column-major, but in a DEVICE
 * buffer of n(n+1)/2 doubles (dnagpu_device_alloc) -- half of the full square a resident dnagpu_matrix takes.  pack: m -> dev_ap on the
 * chain's stream (no staging buffer, no copy stream); unpack: dev_ap -> m (order n, identity padding), stream-ordered as well.
 * (the reference's --staged-adjustment keeps one such packed image per block in its memory-mapped file, dnaadjust-stage.cpp:148-330) */
int dnagpu_matrix_pack_device(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* dev_ap);
int dnagpu_matrix_unpack_device(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* dev_ap, uint32_t n);
int dnagpu_matrix_copy(dnagpu_ctx* ctx, int chain, dnagpu_matrix* dst, const dnagpu_matrix* src);
/* raw copies of a matrix together with its attached junction estimates: np*np doubles (ld = np) followed by np
 * doubles, np = ceil(n/128)*128; dst / src may be host or device memory (this is the payload of the inter-GPU
 * junction exchange, CarryStnEstimatesandVariances* ADJ:998/1133/3196) */
int dnagpu_matrix_export(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* dst, size_t cap_doubles);
int dnagpu_matrix_import(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* src, uint32_t n);
/* The exchange without staging copies: RCCL (ncclBroadcast / ncclSend / ncclRecv) works in place on the matrix's own storage --
 * np * np doubles at *matrix (ld = np) and np doubles at *vector (the attached junction estimates / reduced right-hand side).
 * A receiver sets the logical order first (dnagpu_matrix_resize: n <= n_max; contents untouched). */
int dnagpu_matrix_resize(dnagpu_ctx* ctx, dnagpu_matrix* m, uint32_t n);
int dnagpu_matrix_device_pointers(const dnagpu_matrix* m, double** matrix, double** vector, uint32_t* np);
/* The same for a JUNCTION matrix in either form of dnagpu_schur_carry: np * np + 2 np + 1 doubles -- matrix, attached estimates, the reduced
 * right-hand side of the information form (zeros in the estimates form), the form (0.0 / 1.0); import sets the form it finds.
 * (dna_adjust exchanges v_junctionVariances* / v_junctionEstimates* between its threads by reference, dnaadjust-multi.cpp:365-641;
 * across address spaces they travel as this payload.)  dnagpu_junction_device_pointers: the buffers themselves for an in-place exchange --
 * as_form < 0: a sender, *form tells what it holds (rhs = NULL in the estimates form); as_form 0 / 1: a receiver about to take a
 * junction of that form (the right-hand side's buffer is allocated if need be, the matrix's form is set). */
int dnagpu_junction_export(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* m, double* dst, size_t cap_doubles);
int dnagpu_junction_import(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const double* src, uint32_t n);
int dnagpu_junction_device_pointers(dnagpu_ctx* ctx, dnagpu_matrix* m, int as_form, double** matrix, double** estimates, double** rhs, uint32_t* np,
                                    int* form);
/* Intra-block distributed inverse (networks with fewer blocks than GPUs, e.g. the simultaneous adjustment's one block): `world` contexts,
 * one per GPU, are given the same matrices and call dnagpu_invert together.  Every large launch of the recursion is split by tile
 * columns; a rank computes its columns and `exchange` hands every rank's part to all the others: part q (bufs[q], counts[q] doubles,
 * device memory of this context) is produced by rank q -- the callee broadcasts it from rank q, either enqueued on `stream` (a
 * hipStream_t; RCCL: one ncclBroadcast per part in one group) or completed before it returns.  Returns 0 / nonzero.  Leaves and
 * small launches are computed by every rank, so all contexts end with the same inverse.  world <= 1 or fn == NULL switches it off. */
typedef int (*dnagpu_exchange_fn)(void* user, void* stream, int nparts, double* const* bufs, const size_t* counts);
int dnagpu_set_inverse_exchange(dnagpu_ctx* ctx, int rank, int world, dnagpu_exchange_fn fn, void* user);
/* launches that were split and bytes received through the exchange since the context was created (chain 0) */
int dnagpu_inverse_exchange_stats(dnagpu_ctx* ctx, uint64_t* split_launches, double* bytes_received);
/* in-place inverse (lower in, both triangles out); checks positive definiteness */
int dnagpu_invert(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, int scale_to_unity);

/* ---- blocks ---------------------------------------------------------------- */
/* stations are block-local indices 0..n_stations-1 (ascending global station
 * index, dnaadjust.cpp:10477-10499); unknowns = 3*n_stations. */
int dnagpu_block_create(dnagpu_ctx* ctx, uint32_t blk, uint32_t n_stations, uint32_t n_baselines);
int dnagpu_block_destroy(dnagpu_ctx* ctx, uint32_t blk);
/* xyz: 3*n_stations; sets original == estimated == rigorous */
int dnagpu_block_set_stations(dnagpu_ctx* ctx, uint32_t blk, const double* xyz);
/* the same from DEVICE memory, one launch on the chain's stream, not waited for; with_b: measured-minus-computed of every chain as well
 * (dnagpu_block_compute_b's arithmetic; GNSS-only blocks) -- ResetAdjustment of a network of many small blocks */
int dnagpu_block_reset_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, const double* dev_xyz, int with_b);
/* A set of GNSS-only blocks whose coordinate bookkeeping between iterations is ONE launch for all of them: block blks[q]'s a-priori
 * coordinates at dev_init + init_off[q] (device memory that outlives the table), last[q] != 0 for the last block of a network.
 * apply, mode 0 (ResetAdjustment): original = rigorous = estimated (every chain) = a-priori; mode 1 (UpdateAdjustment, dnaadjust.cpp:496-590):
 * estimated (chains 0 .. chains-1) = rigorous, a last block's original = rigorous; both: measured-minus-computed of those chains
 * (dnagpu_block_compute_b's arithmetic).  Enqueued on the chain's stream, not waited for.  EINVAL for a block with terrestrial measurements. */
/* A run of elimination steps on one chain without a wait per step: while the verdict is held (on != 0) dnagpu_schur_carry and
 * dnagpu_schur_carry_keep return as soon as their launches are enqueued and a pivot that is not positive stays recorded;
 * dnagpu_chain_take_info waits for the chain, returns DNAGPU_ENOTPOSDEF (dnagpu_last_info: the leading minor) if any step of the run failed
 * -- which one is not known: the caller repeats the run with the verdict not held to name it -- and clears the record. */
int dnagpu_chain_hold_info(dnagpu_ctx* ctx, int chain, int on);
int dnagpu_chain_take_info(dnagpu_ctx* ctx, int chain);
typedef struct dnagpu_block_table dnagpu_block_table;
int dnagpu_block_table_create(dnagpu_ctx* ctx, uint32_t n, const uint32_t* blks, const int* last, const double* dev_init, const size_t* init_off,
                              dnagpu_block_table** out);
int dnagpu_block_table_apply(dnagpu_ctx* ctx, int chain, const dnagpu_block_table* t, int mode, int chains);
void dnagpu_block_table_destroy(dnagpu_ctx* ctx, dnagpu_block_table* t);
/* stn1/stn2: block-local station of each baseline (CML order); obs: 3 per
 * baseline (dX,dY,dZ); vcv6: upper triangle per baseline in bms order
 * (XX, XY, YY, XZ, YZ, ZZ), already v-scaled.  Computes W = V^-1 on device. */
int dnagpu_block_set_baselines(dnagpu_ctx* ctx, uint32_t blk, const uint32_t* stn1, const uint32_t* stn2, const double* obs,
                               const double* vcv6);
/* ---- terrestrial measurements (one design row each; types A B C E H K L M R S V Z) ------------------------------
 * The reference evaluates them in UpdateDesignNormalMeasMatrices_A/_BK/_CEM/_E/_M/_S/_V/_Z/_L/_H/_HR/_R
 * (dnaadjust.cpp:4754-6054) every time the design is filled, i.e. at PrepareAdjustment and in every UpdateAdjustment;
 * here dnagpu_block_compute_b does that: computed values, meas-minus-computed, design rows and everything the
 * formation kernels consume.  Call order for a block: create, set_stations, set_station_geo, set_terrestrial,
 * set_clusters (which builds the pair / incidence lists over both kinds of measurement), compute_b. */
/* station records of the block: llh = lat, lon, ellipsoidal height (radians, m; station_t::current*), geoid =
 * geoidSep, defl = 2 per station: verticalDef (prime vertical), meridianDef (radians) */
int dnagpu_block_set_station_geo(dnagpu_ctx* ctx, uint32_t blk, const double* llh, const double* geoid, const double* defl);
/* type[t]; stn3 = station1, station2, station3 (block-local, unused = 0); value = term1 after its one-time reductions;
 * pre_adj_meas = the measurement as supplied (E, M are re-derived from it); variance = term2; instrument / target
 * height = term3 / term4; cml_pos[t] = position of the measurement in the block's CML, cluster_cml_pos[c] likewise
 * for the GNSS clusters given to dnagpu_block_set_clusters afterwards (NULL: cluster c is at position c) */
int dnagpu_block_set_terrestrial(dnagpu_ctx* ctx, uint32_t blk, uint32_t n_t, const char* type, const uint32_t* stn3, const double* value,
                                 const double* pre_adj_meas, const double* variance, const double* inst_height, const double* targ_height,
                                 const uint32_t* cml_pos, const uint32_t* cluster_cml_pos, uint32_t n_clusters);
/* station records <- geodetic coordinates of the chain's current estimates (UpdateGeographicCoords[Phased],
 * dnaadjust.cpp:8711/8734); the design of the next compute_b uses them */
/* Direction sets (type D; UpdateDesignNormalMeasMatrices_D dnaadjust.cpp:5082, LoadVarianceMatrix_D :4059, UpdateAtVinv_D :1328,
 * UpdateNormals_D :1540).  The angles between consecutive directions of a set are entries of type 'D' in
 * dnagpu_block_set_terrestrial (modelled like 'A'), consecutive and with the same cml_pos; instead of a variance of their own
 * they share the set's dense weight matrix (the inverse of the tridiagonal variance matrix of the differences).
 * set_first / set_size: first terrestrial entry and number of angles k of each set; weights: the k x k matrices, column-major,
 * one after the other.
 * Call between dnagpu_block_set_terrestrial and dnagpu_block_set_clusters whenever type 'D' entries exist. */
int dnagpu_block_set_direction_sets(dnagpu_ctx* ctx, uint32_t blk, uint32_t n_sets, const uint32_t* set_first, const uint32_t* set_size,
                                    const double* weights);
int dnagpu_block_update_geodetic(dnagpu_ctx* ctx, int chain, uint32_t blk);
int dnagpu_block_get_station_llh(dnagpu_ctx* ctx, int chain, uint32_t blk, double* llh);
/* meas-minus-computed (n_t) and design rows (9 n_t: dX dY dZ of station 1, 2, 3) of the last compute_b; either may be NULL */
int dnagpu_block_get_terrestrial(dnagpu_ctx* ctx, int chain, uint32_t blk, double* meas_minus_comp, double* design_rows);
/* precision of the adjusted measurements a S a^T (ComputePrecisionAdjMsrs_A/_BCEKLMSVZ/_HIJPQR, dnaadjust.cpp:7877-8007) */
int dnagpu_block_terrestrial_precisions(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_matrix* variances, double* prec);

/* General GNSS measurements: `n_baselines` vectors of 3 design rows each, grouped into clusters that share a dense
 * variance matrix ('G' = cluster of one baseline, 'X' = baseline cluster, 'Y' = point cluster; UpdateDesignNormalMeasMatrices_G
 * / _X / _Y, dnaadjust.cpp:5353 / 6056 / 6249).  stn1[v] = first (negative) station or DNAGPU_NO_STATION for a point,
 * stn2[v] = second (positive) station; cluster c owns vectors cluster_off[c] .. cluster_off[c+1]-1 (CML order);
 * vcv = the clusters' full symmetric 3k x 3k variance matrices (column-major), one after the other, already scaled.
 * W = V^-1 is computed on the device (LoadVarianceMatrix_G/X/Y + FormInverseVarianceMatrix). */
#define DNAGPU_NO_STATION 0xffffffffu
int dnagpu_block_set_clusters(dnagpu_ctx* ctx, uint32_t blk, const uint32_t* stn1, const uint32_t* stn2, const double* obs,
                              uint32_t n_clusters, const uint32_t* cluster_off, const double* vcv);
/* which = 0 original, 1 estimated, 2 rigorous */
int dnagpu_block_get_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, int which, double* xyz);
int dnagpu_block_put_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, int which, const double* xyz);
/* dst <- src within the block (which codes as above) */
int dnagpu_block_copy_stations(dnagpu_ctx* ctx, int chain, uint32_t blk, int dst_which, int src_which);
/* b = obs - (x2 - x1) from the estimated coordinates */
int dnagpu_block_compute_b(dnagpu_ctx* ctx, int chain, uint32_t blk);
int dnagpu_block_get_b(dnagpu_ctx* ctx, int chain, uint32_t blk, double* b);
int dnagpu_block_get_weights(dnagpu_ctx* ctx, int chain, uint32_t blk, double* w6);

/* Post-adjustment statistics of the block's GNSS vectors from the current meas-minus-computed vector (call
 * dnagpu_block_compute_b first) -- ComputePrecisionAdjMsrs_GX/_Y (dnaadjust.cpp:8009/8037) and the per-vector terms of
 * ComputeChiSquare_G/_XY (dnaadjust.cpp:8530/8551):
 *   chi[v]         = b_v . (W b)_v, so that sum_v chi[v] = chi-squared of the block (host sums in CML order);
 *   prec6[6v..6v+5] = upper triangle (xx xy xz yy yz zz) of A S A^T, S = `variances` (the block's rigorous variances,
 *                     3*stations of the block); skipped when `variances` is NULL.  Host buffers. */
int dnagpu_block_msr_statistics(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_matrix* variances, double* prec6, double* chi);

/* m <- sum_i A_i^T W_i A_i (measurement contributions only, CML order) */
int dnagpu_form_normals(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m);
/* m[3s..3s+2, 3s..3s+2] += sign * w9 (column-major 3x3) for k stations */
int dnagpu_add_diag3x3(dnagpu_ctx* ctx, int chain, dnagpu_matrix* m, const uint32_t* stn, const double* w9, size_t k, int sign);
/* rhs = sum_i A_i^T W_i b_i (real measurements) */
int dnagpu_form_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk);
/* ... of up to DNAGPU_BATCH_MAX blocks as two merged launches (the members of a batched condensing step) */
int dnagpu_form_rhs_batched(dnagpu_ctx* ctx, int chain, int nb, const uint32_t* blks);
/* corrections = m * rhs  (m holds N^-1) */
int dnagpu_solve_corrections(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_matrix* m);
/* estimated += corrections; returns the correction of largest magnitude (signed) and its row */
int dnagpu_update_estimates(dnagpu_ctx* ctx, int chain, uint32_t blk, double* max_corr, uint32_t* max_row);
int dnagpu_block_get_corrections(dnagpu_ctx* ctx, int chain, uint32_t blk, double* corr);
/* Oscillation diagnostics (dna_adjust::UpdateIterationDiagnostics, ADJ:7450-7554): per station of the network the correction it was last
 * seen with and its count of consecutive anti-parallel corrections of similar size stay on the device (corrPrev_ / stnOscCount_).
 * dnagpu_osc_reset: n_stations records, all unseen (start of an adjustment).  dnagpu_osc_block: one block's visit of this iteration --
 * its corrections of chain `corr_chain` against the records of its stations (`stations`: their indices in the network, kept after the
 * first call), on chain 0's stream; call the blocks in order.  dnagpu_osc_flagged: how many visits since the last call found a count of
 * 2 or more (and resets that counter; synchronises).  dnagpu_osc_block_visits: per station of the block the count where it is >= 2, else 0.
 * A block's corrections can be set aside (dnagpu_block_keep_corrections: a copy on the chain's stream) where a later solve of the same
 * block on the same chain would overwrite them -- the reference restores the last block's forward corrections after its reverse solve
 * (UpdateEstimatesFinal, ADJ:3755); corr_chain / chain = -1 names the copy in dnagpu_osc_block / dnagpu_block_get_corrections. */
int dnagpu_block_keep_corrections(dnagpu_ctx* ctx, int chain, uint32_t blk);
int dnagpu_osc_reset(dnagpu_ctx* ctx, size_t n_stations);
int dnagpu_osc_block(dnagpu_ctx* ctx, uint32_t blk, int corr_chain, const uint32_t* stations);
/* ... and the visits of n blocks, in the order given, as ONE launch (the blocks follow each other inside it: a station shared by several blocks is
 * compared with its previous visit, which may be an earlier block's of this iteration) */
int dnagpu_osc_blocks(dnagpu_ctx* ctx, uint32_t n, const uint32_t* blks, const int* corr_chain, const uint32_t* const* stations);
int dnagpu_osc_flagged(dnagpu_ctx* ctx, uint32_t* n_flagged);
int dnagpu_osc_block_visits(dnagpu_ctx* ctx, uint32_t blk, uint32_t* visit);
int dnagpu_block_get_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, double* rhs);

/* ---- junction carry -------------------------------------------------------- */
/* A junction set = k stations; idx_from / idx_to are their block-local station
 * indices in the source and destination block. */
/* jm (order 3k) <- rows/cols of src (N^-1) at stations idx; jest (3k, device vector
 * owned by the junction matrix) <- estimated coordinates of those stations.
 * src == NULL: only the estimates are refreshed, the matrix already held by jm stays (it does not change between the
 * iterations of a GNSS-only network) */
int dnagpu_junction_gather(dnagpu_ctx* ctx, int chain, uint32_t blk_from, const dnagpu_matrix* src, const uint32_t* idx_from,
                           size_t k, dnagpu_matrix* jm);
/* The carry of a forward / reverse step WITHOUT the block inverse.  The reference solves the whole block
 * (dna_adjust::Solve, dnaadjust.cpp:6586: n^3 flops), gathers the junction block of N^-1, inverts it
 * (CarryStnEstimatesandVariancesForward / ...Reverse, dnaadjust.cpp:998-1128, 1133-1281) and carries that weight matrix
 * plus the junction estimates; nothing else of a forward / reverse solution is used unless the block is the last / first
 * of its network.  ((N^-1)_JJ)^-1 is the Schur complement  N_JJ - N_JI N_II^-1 N_IJ  and the junction corrections solve
 * S dx_J = rhs_J - N_JI N_II^-1 rhs_I : both come out of eliminating the inner unknowns only (~0.35 n^3 flops).
 * m: the block's normals (lower triangle; contents are destroyed), rhs as left by dnagpu_form_rhs / dnagpu_junction_rhs.
 * jm (order 3k) <- S, and
 *   estimates form (DNAGPU_INFO_CARRY=0 / dnagpu_debug_set_info_carry(0)): jest <- estimated coordinates + corrections of the k
 *     listed stations: exactly what dnagpu_junction_gather + dnagpu_invert leave there after a full solve, up to rounding;
 *   information form (default, round 4): jest <- the estimated coordinates the block was linearised at, and the reduced right-hand
 *     side r = rhs_J - N_JI N_II^-1 rhs_I beside it.  dnagpu_junction_rhs adds  r + S (jest - the receiving block's estimates)  --
 *     what the weighted pseudo-observation "estimates + S^-1 r" of the estimates form contributes, with S^-1 cancelled
 *     analytically: the complement is never factored or inverted (nj^3 flops less per step; for the 450-unknown junctions of
 *     a dnasegment-like cut, half of a chain step).  Such a matrix travels by dnagpu_junction_export / _import /
 *     _device_pointers (dnagpu_matrix_export and dnagpu_matrix_device_pointers, which know nothing of its right-hand side, refuse it).
 * The block's estimates and corrections are NOT updated.  DNAGPU_ENOTPOSDEF like dnagpu_invert. */
int dnagpu_schur_carry(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m, const uint32_t* idx_out, size_t k, dnagpu_matrix* jm);
/* The carry of a chain step whose system does not change between iterations (GNSS-only network, a.reuse_factors): the same elimination,
 * information form only, with its factor KEPT in `keep` (a light factor with storage of its own: dnagpu_partial_create_spine(...,
 * store = NULL, ...) for the step's n and k).  In every later iteration dnagpu_schur_carry_rhs takes the step's right-hand side
 * (rhs(blk) as left by dnagpu_block_load_reduced / dnagpu_junction_rhs) through the kept factor by blocked substitution: jm's reduced
 * right-hand side and linearisation point are renewed, its matrix (the complement S) stays -- no factorisation, HBM-bound.
 * Replaces the repeated Solve() + CarryStnEstimatesandVariances* of dnaadjust.cpp:2812 / 998-1281 for iterations >= 2, the way the
 * reference itself skips the inverse after iteration 1 in simultaneous mode (dnaadjust.cpp:2452-2457). */
int dnagpu_schur_carry_keep(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m, const uint32_t* idx_out, size_t k, dnagpu_matrix* jm,
                            dnagpu_partial* keep);
int dnagpu_schur_carry_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, const uint32_t* idx_out, size_t k, dnagpu_matrix* jm, const dnagpu_partial* keep);
/* A whole chain step of that kind on a SMALL condensed system as ONE launch (small_steps.hip): dnagpu_block_load_reduced(rblk, src_blk,
 * idx_keep, red, m = NULL) + dnagpu_junction_rhs(rblk, idx_in, jm_in) (jm_in = NULL: no junction carried in) + dnagpu_schur_carry_rhs(rblk,
 * idx_out, jm_out, keep) -- vectors in LDS, the kept factor streamed once, nothing waits for the host.  Both junctions in information
 * form.  DNAGPU_ETOOLARGE (and nothing done) when the padded system or the junction carried in exceeds 2 048 unknowns or an index list
 * could not be cached: the caller then issues the three calls.  dnasegment's default cut (150 stations per block, dnaoptions.hpp:382:
 * condensed systems of a few hundred unknowns, hundreds of steps in a row) is what this is for. */
int dnagpu_chain_step_rhs(dnagpu_ctx* ctx, int chain, uint32_t rblk, uint32_t src_blk, const uint32_t* idx_keep, size_t k_keep, const dnagpu_matrix* red,
                          const dnagpu_matrix* jm_in, const uint32_t* idx_in, size_t k_in, dnagpu_matrix* jm_out, const uint32_t* idx_out, size_t k_out,
                          const dnagpu_partial* keep);
/* The per-block steps of an iteration >= 2 (a.reuse_factors) for MANY small blocks as one launch each, a workgroup per block (small_steps.hip):
 *   dnagpu_small_batch_condense   dnagpu_form_rhs + dnagpu_partial_reduce_rhs of every block of the batch (chain 0's vectors of the blocks)
 *   dnagpu_small_batch_solve      every block's rigorous solve: estimates <- originals (not for a `last` block), right-hand side, + r + S dx of
 *                                 the junctions j0 then j1 (information form; NULL: none; idx: their stations in the block), substitution with
 *                                 the completed light factor pf, estimates += corrections, rigorous <- estimates, originals <- rigorous (not
 *                                 for a last block, whose corrections are set aside like dnagpu_block_keep_corrections); max_corr[q] <- the
 *                                 block's correction of largest magnitude.  What PhasedForwardBlock / ...ReverseBlock / ...CombineBlock +
 *                                 UpdateEstimates* + UpdateEstimatesFinal do per block (dnaadjust.cpp:2812-3057, 3512-3800) for iterations >= 2.
 * create: DNAGPU_ETOOLARGE (nothing made) when a block is beyond the one-workgroup kernels (padded order or junction > 2 048 unknowns,
 * terrestrial rows, a factor that is not a completed light one): the caller keeps the per-block calls.  The batch refers to the blocks',
 * factors' and junction matrices' storage: destroy it before any of them. */
typedef struct dnagpu_small_batch dnagpu_small_batch;
int dnagpu_small_batch_create(dnagpu_ctx* ctx, uint32_t n, const uint32_t* blks, dnagpu_partial* const* pf, dnagpu_matrix* const* red,
                              const dnagpu_matrix* const* j0, const uint32_t* const* idx0, const size_t* k0, const dnagpu_matrix* const* j1,
                              const uint32_t* const* idx1, const size_t* k1, const int* last, dnagpu_small_batch** out);
int dnagpu_small_batch_condense(dnagpu_ctx* ctx, int chain, dnagpu_small_batch* sb);
int dnagpu_small_batch_solve(dnagpu_ctx* ctx, int chain, dnagpu_small_batch* sb, double* max_corr);
void dnagpu_small_batch_destroy(dnagpu_ctx* ctx, dnagpu_small_batch* sb);
/* ---- chain plans: the chain steps of the condensed schedule as data, taken through the elimination in lock-step batches ----
 * A chain step -- PhasedForwardBlock / PhasedReverseBlock on a condensed block (dnaadjust.cpp:2812, 3109, with
 * CarryStnEstimatesandVariancesForward / ...Reverse :998-1281), a merge of two condensed systems, a step over a run's system -- adds a
 * few systems into one (reduced systems: matrix + right-hand side in the attached vector; junction matrices in information form),
 * adds constraint blocks, eliminates all stations but `keep` and leaves their complement in `out` (an information-form junction with
 * the kept stations' estimates, or a reduced system).  dnagpu_chain_plan_create puts n_steps such steps on the device, grouped into
 * batches (batch q = steps batch_first[q] .. batch_first[q + 1] - 1, at most DNAGPU_CHAIN_BATCH_MAX -- DNAGPU_BATCH_MAX for matrix_only steps --, independent of each other; its members are
 * eliminated in one padded shape, the largest of theirs) and owns their factors; dnagpu_chain_plan_run takes one batch through assembly,
 * elimination (factor kept, light form) and output as ONE sequence of merged launches, without waiting (the verdict of the eliminations
 * stays in the chain: dnagpu_chain_hold_info / _take_info); dnagpu_chain_plan_run_rhs takes the steps of batches q_lo .. q_hi - 1
 * (independent of each other, factored before) through their kept factors, right-hand sides only, in ONE launch.  A long chain of
 * small steps -- a dnasegment-default cut -- is bound by its length: cut into runs that advance together it is 2 B / W + W steps deep
 * instead of B (dna_adjust::LockstepChains).  DNAGPU_ETOOLARGE (nothing made) when a step is beyond the small-system kernels
 * (more than 2 048 unknowns).  Factors beyond `max_bytes`: the plan keeps none (dnagpu_chain_plan_info).  The plan refers to the matrices' and
 * blocks' storage: destroy it before any of them. */
typedef struct dnagpu_chain_plan dnagpu_chain_plan;
typedef struct dnagpu_chain_source {
    const dnagpu_matrix* m;
    int junction;               /* 0: a reduced system (right-hand side in the attached vector); 1: a junction matrix, information form */
    const uint32_t* pos;        /* station a of m is station pos[a] of the step's system */
    size_t k;
} dnagpu_chain_source;
typedef struct dnagpu_chain_step {
    uint32_t n_stn;                                   /* stations of the step's system */
    const uint32_t* est_blk; const uint32_t* est_idx; /* the linearisation point: station s = station est_idx[s] of block est_blk[s] (its originals); NULL: none */
    int n_src;
    dnagpu_chain_source src[3];
    const uint32_t* con_stn; const double* con_w9; size_t n_con;      /* constraint blocks (dnagpu_add_diag3x3) */
    const uint32_t* keep; size_t n_keep;              /* the stations carried on, in the order of out's unknowns */
    dnagpu_matrix* out;
    int out_junction;                                 /* 1: information-form junction (estimates of the kept stations attached); 0: reduced system */
    int matrix_only;                                  /* 1: nothing is eliminated and nothing carried on (keep, out, est_*: unused): the assembled MATRIX is the kept
                                                         block of a block's rigorous solve, taken by dnagpu_partial_complete_factor_planned (a batch holds
                                                         steps of one kind only) */
} dnagpu_chain_step;
int dnagpu_chain_plan_create(dnagpu_ctx* ctx, size_t n_steps, const dnagpu_chain_step* steps, size_t n_batches, const uint32_t* batch_first,
                             double max_bytes, dnagpu_chain_plan** out);
/* keeps_factors: 0 when the steps' factors exceeded max_bytes -- the plan then keeps none (dnagpu_chain_plan_run eliminates into scratch of its chain
 * every time, dnagpu_chain_plan_run_rhs is refused); factor_bytes: what the kept factors take */
int dnagpu_chain_plan_info(const dnagpu_chain_plan* plan, int* keeps_factors, double* factor_bytes);
int dnagpu_chain_plan_run(dnagpu_ctx* ctx, int chain, dnagpu_chain_plan* plan, size_t batch);
int dnagpu_chain_plan_run_rhs(dnagpu_ctx* ctx, int chain, dnagpu_chain_plan* plan, size_t batch_lo, size_t batch_hi);
void dnagpu_chain_plan_destroy(dnagpu_ctx* ctx, dnagpu_chain_plan* plan);
/* dnagpu_partial_complete_factor_batched with the members' kept blocks described by the matrix_only steps of batch `batch` (step b <-> pf[b]):
 * the kept blocks -- reduced block + carried junction weights + constraints, what dna_adjust::PrepareKeptBlock puts together call by call --
 * are assembled by ONE launch straight into the members' matrices, then factored and inverted in lock step. */
int dnagpu_partial_complete_factor_planned(dnagpu_ctx* ctx, int chain, dnagpu_chain_plan* plan, size_t batch, dnagpu_partial* const* pf, int* failed_member);
/* The same elimination as a stand-alone step: red (order 3k) <- Schur complement of all other unknowns of m onto the k
 * listed stations (list order), red's attached vector <- the reduced right-hand side.  With the stations a block shares
 * with its neighbours as the list, this condenses the block to its junction stations ONCE per iteration, independently of
 * every other block; the forward and the reverse chain of dna_adjust::AdjustPhased then run on the condensed blocks
 * (a few thousand unknowns each) and produce the very junction weights and estimates the block-level chain would. */
int dnagpu_block_reduce(dnagpu_ctx* ctx, int chain, uint32_t blk, dnagpu_matrix* m, const uint32_t* idx_keep, size_t k, dnagpu_matrix* red,
                        dnagpu_partial* keep /* may be NULL */);
/* The same for a block whose factor is kept (`keep` required), WITHOUT a formed matrix to start from: the normals of the block's current
 * estimates (what dnagpu_form_normals builds) plus the listed constraint blocks (dnagpu_add_diag3x3, sign +1) are formed directly in the
 * unknown order the elimination works in, the right-hand side (dnagpu_form_rhs before this call) as its passenger row -- one pass over
 * the matrix less than form + add + reduce.  Same bits as that sequence. */
int dnagpu_block_form_reduce(dnagpu_ctx* ctx, int chain, uint32_t blk, const uint32_t* con_stn, const double* con_w9, size_t n_con,
                             const uint32_t* idx_keep, size_t k, dnagpu_matrix* red, dnagpu_partial* keep);
/* A light factor (dnagpu_partial_create_spine, after its reduce) as the packed lower triangle of its npp x npp storage,
 * npp (npp + 1) / 2 doubles of device memory (npp <= the capacity: n_max + 256 covers it), and back: dst takes src's orders and
 * unknown order with the factor (dst == src allowed) and is then what src was after its reduce.  Exact copies. */
int dnagpu_partial_pack_device(dnagpu_ctx* ctx, int chain, const dnagpu_partial* p, double* dev_ap);
int dnagpu_partial_unpack_device(dnagpu_ctx* ctx, int chain, dnagpu_partial* dst, const dnagpu_partial* src, const double* dev_ap);
/* ... and through page-locked HOST memory (round 6): a block whose packed variance matrix will live in a host slot of the staged store parks its
 * factor there -- packed on the chain's stream, copied on the chain's copy stream (dnagpu_copies_sync before the slot is read or written again);
 * back: copied and unpacked on the chain's stream.  host_ap: (npp (npp + 1) / 2) doubles, npp <= n + 256.  Replaces the second and third
 * dpotrf of a block per iteration that the reference's Solve() repeats anyway (dnaadjust.cpp:6586-6647). */
int dnagpu_partial_pack_host_async(dnagpu_ctx* ctx, int chain, const dnagpu_partial* p, double* host_ap);
int dnagpu_partial_unpack_host(dnagpu_ctx* ctx, int chain, dnagpu_partial* dst, const dnagpu_partial* src, const double* host_ap);
/* A chain's inverse workspace (X and W: two (n_max + 256)^2 matrices, padded) is allocated on the chain's first call that needs it;
 * dnagpu_chain_reserve makes that allocation now -- in PrepareAdjustment rather than inside the first iteration (hipMalloc of
 * 2 x 5.9 GB per chain at n = 27 000 is seconds). */
int dnagpu_chain_reserve(dnagpu_ctx* ctx, int chain, uint32_t n_max);
/* ... and the chain's copy stream with its staging buffer of `doubles` (dnagpu_matrix_download_packed_async, dnagpu_partial_pack_host_async,
 * dnagpu_partial_unpack_host), so that the staged store's first copy does not allocate either. */
int dnagpu_copy_stage_reserve(dnagpu_ctx* ctx, int chain, size_t doubles);
/* Batched forms of the three large steps of a block with a kept factor in its light form (dnagpu_partial_create_spine), for nb <=
 * DNAGPU_BATCH_MAX blocks of ONE shape (equal padded orders of the eliminated and of the kept part): the members' launches are merged --
 * every tile product and every leaf of the recursion is one launch that works on all members, in lock step.  The dependent chain of
 * launches is as long as for one block, but every launch has nb times the tiles: the short launches at the bottom of the recursion
 * fill the GPU, and the large ones follow each other without a partly filled last wave (replaces nb x dpotrf / dpotri call sequences of
 * matrix_2d::cholesky_inverse, dnamatrix_contiguous.cpp:982-1006, that the reference runs one after the other).  Same bits per member
 * as the unbatched calls.  A member that is not positive definite: DNAGPU_ENOTPOSDEF, *failed_member = its position (may be NULL).
 * dnagpu_batch_reserve allocates the members' workspaces on `chain` (a matrix + the panels of a diagonal block each) for nb_wanted blocks of
 * n_max unknowns with k_max kept ones: *nb_granted = nb_wanted, or 1 when they do not fit (nothing stays allocated then beside the chain's
 * own workspace; the caller runs the blocks one at a time).  The batched calls allocate the same on demand and fail with DNAGPU_ENOMEM. */
#define DNAGPU_BATCH_MAX 32
#define DNAGPU_CHAIN_BATCH_MAX 32   /* steps per batch of a chain plan (dnagpu_chain_plan_create) */
int dnagpu_batch_reserve(dnagpu_ctx* ctx, int chain, uint32_t n_max, uint32_t k_max, int nb_wanted, int* nb_granted);
int dnagpu_block_form_reduce_batched(dnagpu_ctx* ctx, int chain, int nb, const uint32_t* blks, const uint32_t* const* con_stn,
                                     const double* const* con_w9, const size_t* n_con, const uint32_t* const* idx_keep, const size_t* k,
                                     dnagpu_matrix* const* red, dnagpu_partial* const* keep, int* failed_member);
int dnagpu_partial_complete_factor_batched(dnagpu_ctx* ctx, int chain, int nb, dnagpu_partial* const* pf, const dnagpu_matrix* const* kk,
                                           int* failed_member);
int dnagpu_partial_finish_batched(dnagpu_ctx* ctx, int chain, int nb, dnagpu_partial* const* pf, dnagpu_matrix* const* inv);
/* With `keep`, the elimination leaves everything a later completion needs in HBM -- the factor of the eliminated part, its
 * inverse and the panel under the kept rows (n^2 + 3k n doubles) -- at 2/3 n_i^3 instead of ~0.34 n_i^3 flops.
 * dnagpu_partial_complete then turns  [ N_II  . ; N_KI  kk ]  (kk: the kept block as the chains left it: reduced block +
 * carried junction weights + constraints, order 3k, the list order of the reduce) into its full inverse `inv`, in the block's
 * natural unknown order, for n^3/3 + O(n_i^2 k) flops -- instead of forming the block's normals again and inverting them
 * (dna_adjust::Solve, n^3).  The retained state is consumed. */
int dnagpu_partial_create(dnagpu_ctx* ctx, uint32_t n_max, uint32_t k_max, dnagpu_partial** out);
/* The same without storage of its own for the factor's inverse: between the elimination and the completion it lives in `store` -- a
 * matrix created with n_max + 256 whose content is void meanwhile (typically the block's rigorous variance matrix, which is dead from
 * the start of an iteration until the completion writes it: `inv` of dnagpu_partial_complete may be `store` itself).  Costs the panel
 * (3k n doubles) and nothing else; dnagpu_partial_reduce_rhs is not available afterwards.  `store` must outlive the partial. */
int dnagpu_partial_create_in(dnagpu_ctx* ctx, uint32_t n_max, uint32_t k_max, dnagpu_matrix* store, dnagpu_partial** out);
void dnagpu_partial_destroy(dnagpu_ctx* ctx, dnagpu_partial* p);
/* The light form: the elimination stops at the factor (~0.34 n_i^3 instead of 2/3 n_i^3 -- no inverse of the eliminated part is formed), kept
 * in `store` as one block lower triangular matrix (inverses of the diagonal blocks, panels below them).  dnagpu_partial_complete_factor then
 * only factors the kept block, dnagpu_partial_solve substitutes block by block, and dnagpu_partial_finish pays for the inverse of the
 * factor as well (2/3 n^3 in all) -- once, after the last iteration.  No panel copy; `store` = NULL gives it storage of its own (n^2).
 * The capacity is the SHAPE the block is eliminated in: the eliminated part padded (with an identity) to ceil128(n_max - k_max), the kept
 * part to ceil128(k_max + 1).  A block smaller than its factor's capacity is simply padded further -- blocks of unequal size given one
 * common capacity thereby share a shape and can go through the batched calls together, each with the bits it would get alone. */
int dnagpu_partial_create_spine(dnagpu_ctx* ctx, uint32_t n_max, uint32_t k_max, dnagpu_matrix* store, dnagpu_partial** out);
int dnagpu_partial_complete(dnagpu_ctx* ctx, int chain, dnagpu_partial* pf, const dnagpu_matrix* kk, dnagpu_matrix* inv);
/* The completion in two halves.  An iteration of the adjustment needs the block's solution, not its inverse: the reference obtains the
 * one through the other (dna_adjust::Solve: N^-1, then N^-1 rhs), and only the inverse of the LAST iteration is a result (the rigorous
 * variances).  dnagpu_partial_complete_factor turns the kept state into X = L^-1 of the whole block (the kept block factored and
 * inverted, the two panel products: ~0.08 n^3); dnagpu_partial_solve gives  corrections(blk) = X^T (X rhs(blk))  -- two triangular
 * matrix-vector products, what dnagpu_solve_corrections computes from the inverse --; dnagpu_partial_finish, once the iterations
 * have ended, forms  inv = X^T X  (n^3 / 3) in the block's natural order.  factor + finish == dnagpu_partial_complete. */
int dnagpu_partial_complete_factor(dnagpu_ctx* ctx, int chain, dnagpu_partial* pf, const dnagpu_matrix* kk);
int dnagpu_partial_solve(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_partial* pf);
int dnagpu_partial_finish(dnagpu_ctx* ctx, int chain, dnagpu_partial* pf, dnagpu_matrix* inv);
/* After a completion the factor of the eliminated part is still there.  When the block's normals do not change between
 * iterations (GNSS-only network), the reduced right-hand side of the next iteration is  rhs_K - L_KI (L_II^-1 rhs_I) : two
 * matrix-vector products with the kept X = L_II^-1 and the kept panel instead of a new elimination.  red's vector <- that; red's
 * matrix (the Schur complement) is left as it is.  A light factor (dnagpu_partial_create_spine) serves from its reduce on, completed
 * (dnagpu_partial_complete_factor) or not, until dnagpu_partial_finish: the forward half of dnagpu_partial_solve's substitution. */
int dnagpu_partial_reduce_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, const dnagpu_partial* pf, dnagpu_matrix* red);
/* Start a chain step on a condensed block: m <- red (m = NULL: not needed, the step's factor is kept), rhs(rblk) <- red's vector,
 * estimated(rblk) <- original(src_blk)[idx_keep].  rblk: a block created with k stations and no measurements. */
int dnagpu_block_load_reduced(dnagpu_ctx* ctx, int chain, uint32_t rblk, uint32_t src_blk, const uint32_t* idx_keep, size_t k,
                              const dnagpu_matrix* red, dnagpu_matrix* m);
/* dst[idx,idx] += jm (3x3 blocks), and rhs_extra of blk_to gets the pseudo
 * measurement part:  rhs[idx] += jm * (jest - estimated_to[idx])  is applied
 * by dnagpu_junction_rhs at solve time. */
int dnagpu_junction_scatter(dnagpu_ctx* ctx, int chain, dnagpu_matrix* dst, const uint32_t* idx_to, size_t k,
                            const dnagpu_matrix* jm);
/* rhs(blk_to)[idx] += jm * (jest(jm) - estimated(blk_to)[idx]) */
int dnagpu_junction_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk_to, const uint32_t* idx_to, size_t k, const dnagpu_matrix* jm);
/* Condensed systems as normal-equation pairs (matrix, right-hand side), for the two-level chain across GPUs (a run of condensed
 * blocks reduced to the stations of its two ends):
 *   dnagpu_block_add_rhs          rhs(blk)[idx] (+)= the vector attached to jm (3k doubles: a reduced right-hand side), optionally
 *                                 after zeroing the whole right-hand side of blk; with dnagpu_matrix_reset + dnagpu_junction_scatter
 *                                 this assembles two overlapping condensed systems into one, which dnagpu_block_reduce condenses again
 *   dnagpu_block_gather_stations  estimated(dst_blk)[dst_pos] <- original(src_blk)[src_idx]: the linearisation point of a
 *                                 station-less block that stands for such a system (what dnagpu_block_load_reduced does from one source) */
int dnagpu_block_add_rhs(dnagpu_ctx* ctx, int chain, uint32_t blk, const uint32_t* idx, size_t k, const dnagpu_matrix* jm, int zero_first);
int dnagpu_block_gather_stations(dnagpu_ctx* ctx, int chain, uint32_t dst_blk, const uint32_t* dst_pos, uint32_t src_blk, const uint32_t* src_idx,
                                 size_t k);
/* read / write the junction estimates attached to a junction matrix (3k doubles).  After dnagpu_schur_carry in its information form
 * (the default) "the estimates" are those the block was linearised at, not adjusted ones; writing estimates makes the matrix one in the
 * estimates form (its reduced right-hand side is dropped). */
int dnagpu_junction_get_estimates(dnagpu_ctx* ctx, int chain, const dnagpu_matrix* jm, double* est);
int dnagpu_junction_put_estimates(dnagpu_ctx* ctx, int chain, dnagpu_matrix* jm, const double* est, size_t k);

/* ---- stream ordering between chains ---------------------------------------- */
/* make `waiter` chain wait for everything enqueued so far on `signaller` */
int dnagpu_chain_wait(dnagpu_ctx* ctx, int waiter, int signaller);
int dnagpu_chain_sync(dnagpu_ctx* ctx, int chain);

#ifdef __cplusplus
}
#endif
#endif /* DNAGPU_H_ */
