// Host driver for the in-place symmetric positive definite inverse on the device.
// Restates matrix_2d::cholesky_inverse (dnamatrix_contiguous.cpp:952-1020:
// dpotrf + dpotri) and the optional diagonal scaling of dna_adjust::Solve
// (dnaadjust.cpp:6614-6645) as a recursive sequence of tile-GEMM launches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <set>
#include <utility>
#include <vector>
#include "la_kernels.h"

namespace dnagpu {

struct GemmProfile {
    bool enabled = false;
    double flops = 0.0;      // actual flops issued by gemm launches since reset
    double gemm_ms = 0.0;    // summed event time of gemm launches (after collect())
    uint64_t launches = 0;
    std::vector<hipEvent_t> pool;  // start/stop pairs, one pair per run of consecutive gemm launches
    size_t used = 0;
    bool open = false;       // a run is in progress (start event recorded, stop event pending)
};

// A batched driver call: the drivers below run on member 0's buffers as always, and every launch they enqueue carries nb members --
// the same product or leaf on nb matrices of one shape, in lock step (la_kernels.h: GemmArgs::nb).  The buffers a call touches are
// registered here (member 0's address range, the other members' offsets in doubles); gemm() finds an operand's buffer by its
// address.  While a batch is set the split-launch path is off, and P is addressed relative to the diagonal block being factored
// (Rec::p_o) so that the members' panel buffers only need that block's columns.
struct InvBatch {
    int nb = 1;
    int nbuf = 0;
    const double* base[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t span[4] = {0, 0, 0, 0};
    long long delta[4][BATCH_MAX];
    void add(const double* member0, size_t doubles, double* const* members) {
        base[nbuf] = member0;
        span[nbuf] = doubles;
        for (int b = 0; b < nb; ++b) delta[nbuf][b] = (long long)(members[b] - member0);
        ++nbuf;
    }
};

// Workspace shared by every inverse run on one stream ("chain").
struct InvWorkspace {
    double* X = nullptr;     // np_cap^2 : L^-1
    double* W = nullptr;     // np_cap^2 : L21 panels
    double* svec = nullptr;  // np_cap   : diagonal scaling
    int* info = nullptr;     // device ints (dpotrf-style info, 0 = ok), one per member of a batched call (BATCH_MAX)
    int* info_host = nullptr;  // pinned host copy
    bool hold_info = false;    // dnagpu_chain_hold_info: the drivers leave `info` alone (the first failure of a run of calls stays in it)
    InvBatch batch;          // nb > 1: the call being enqueued is batched
    double* bX[BATCH_MAX] = {};   // members 1 .. of a batched call: their matrix being factored (bnp_cap^2) ...
    double* bW[BATCH_MAX] = {};   // ... and the panels inside a diagonal block (bw_cols x bnp_cap)
    uint32_t bnp_cap = 0, bw_cols = 0;
    uint64_t batched_launches = 0;
    uint32_t np_cap = 0;
    hipStream_t stream = nullptr;
    GemmProfile prof;
    // workgroup->tile tables per launch shape (see tile_order.hip) and tile-column range (0xffffffff: all columns), device resident
    std::map<std::pair<uint64_t, uint32_t>, std::pair<uint32_t*, int>> order_cache;
    std::set<uint64_t> planned;  // driver call shapes (sym_inverse.hip plan_key) whose tile-order tables are all built
    // First HIP error of a planning pass, table upload, memset / copy or kernel launch enqueued through this workspace since
    // the last inv_take_error(): the asynchronous drivers below keep enqueuing nothing further once it is set, and the C-ABI
    // turns it into DNAGPU_ENOMEM / DNAGPU_EHIP instead of trusting `info_host` (a skipped GEMM leaves info at "no failure").
    hipError_t err = hipSuccess;
    const char* err_where = nullptr;
    // Intra-block distributed inverse (dnagpu_set_inverse_exchange): `dist_world` GPUs hold the same matrices; every large launch is
    // split by tile columns, each rank computes its columns, `exchange` makes every rank's part known to all (part q comes from rank
    // q; enqueued on, or completed before returning to, the given stream).  Leaves and small launches are done by everybody.
    int dist_rank = 0, dist_world = 1;
    int (*exchange)(void* user, void* stream, int nparts, double* const* bufs, const size_t* counts) = nullptr;
    void* exchange_user = nullptr;
    double* dist_stage = nullptr;
    size_t dist_stage_cap = 0;
    uint64_t split_launches = 0;
    double exchanged_bytes = 0.0;
};

// returns the latched error (and where it happened) and clears it
hipError_t inv_take_error(InvWorkspace& ws, const char** where);
// latches e (if it is the first) -- also used by the callers for their own enqueues on ws.stream
void inv_note_error(InvWorkspace& ws, hipError_t e, const char* where);

struct GemmArgs;
// fills a.order / a.grid from the cache (building + uploading the table on first use)
hipError_t gemm_attach_order(InvWorkspace& ws, GemmArgs& a, int jt_lo = -1, int jt_hi = -1);

// returns hipSuccess or an error; allocates for matrices up to np_cap (multiple of 128)
hipError_t inv_workspace_alloc(InvWorkspace& ws, uint32_t np_cap, hipStream_t stream);
void inv_workspace_free(InvWorkspace& ws);

// F: np x np (ld = np) device buffer, lower triangle valid, identity padding beyond n.
// On return F holds the inverse in BOTH triangles.  Asynchronous on ws.stream; the
// dpotrf-style info is copied to ws.info_host (valid after stream sync).
void sym_inverse_async(InvWorkspace& ws, double* F, uint32_t n, uint32_t np, bool scale_to_unity, bool reset_info = true);

// Partial elimination for the junction carry (dnagpu_schur_carry).  F: (ti + tj) x (ti + tj) tiles, ld, lower tiles valid;
// the leading ti tile rows / columns are eliminated (Cholesky panels into P, ldp; the inverses of the diagonal blocks
// into ws.X) and the trailing tj x tj tiles of F become the Schur complement  F22 - F21 F11^-1 F12  (lower tiles).
// Resets ws.info; a non-positive pivot in the eliminated part is reported like dpotrf.  The caller copies ws.info back.
void sym_schur_async(InvWorkspace& ws, double* F, int ld, double* P, int ldp, int ti, int tj);

// The same elimination with everything kept that a later completion to the full inverse needs (dnagpu_partial): the whole
// leading part is factored AND inverted (F: T21 pieces, X: L_II^-1, both ld), the panel L_KI of the trailing rows is left in
// ws.W (rows ti*128.., ld) for the caller to save, the trailing tiles of F become the Schur complement.  2/3 n_i^3 instead of
// ~0.34 n_i^3 flops -- the completion then costs n^3/3 + O(n_i^2 n_k) instead of a new n^3 inverse.
void sym_schur_keep_async(InvWorkspace& ws, double* F, double* X, int ld, int ti, int tj);
// F: trailing tj x tj tiles hold the (updated) kept block; X, F leading parts as sym_schur_keep_async left them; WK: the saved
// panel L_KI (tj*128 x ti*128, ldwk).  On return F = inverse of the whole matrix, both triangles, in the elimination's order.
// what: 1 = the factor only (X = L^-1 of the whole matrix complete; F is scratch), 2 = the inverse X^T X -> F only (after a 1), 3 = both
void sym_complete_async(InvWorkspace& ws, double* F, double* X, int ld, const double* WK, int ldwk, int ti, int tj, int what = 3);

// The kept factor in its light form (a.defer_variances = 2): sym_spine_async eliminates like sym_schur_async (~0.34 n_i^3) but leaves the
// factor in S (ld) -- inverses of the diagonal blocks of the elimination's block sequence (sym_spine_blocks) on the diagonal, the panels
// L_(below, b) below them; F's trailing tiles become the Schur complement.  sym_spine_kept_async factors and inverts the (updated)
// trailing tj x tj block of F into S's trailing block: S then solves any right-hand side by blocked substitution.  sym_spine_finish_async
// turns S into L^-1 (n^3 / 3) and F into the inverse X^T X (n^3 / 3), both triangles.
void sym_spine_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj);
void sym_spine_kept_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj);
void sym_spine_finish_async(InvWorkspace& ws, double* F, double* S, int ld, int ti, int tj);
std::vector<std::pair<int, int>> sym_spine_blocks(int ti);      // (first tile, tiles) of the diagonal blocks, in elimination order

// sum the event timings recorded so far (synchronises the stream)
void gemm_profile_collect(InvWorkspace& ws);
void gemm_profile_close(InvWorkspace& ws);   // ends the current run of gemm launches (call before enqueuing any other kernel)
void gemm_profile_reset(InvWorkspace& ws);

double schur_split();

// fault_inject_reset(n): the n-th tile-table allocation from now on fails with hipErrorOutOfMemory (tests/test_gpu_matrix.py:
// a failed allocation in the middle of an inverse must surface as DNAGPU_ENOMEM, never as a silently skipped launch)
void fault_inject_reset(long nth);
// threshold (in 128-tiles per launch) below which a launch uses the 64-tile kernel; negative restores the default; returns the old value
long small_tiles_set(long v);
long tiny_tiles_set(long v);      // 128-tiles per launch below which the product runs on 32 x 32 tiles (< 0: the default, TINY_LAUNCH_TILES)

}  // namespace dnagpu
