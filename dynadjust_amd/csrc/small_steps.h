// Whole steps of the adjustment on small systems as one launch of one workgroup (small_steps.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnagpu {

constexpr uint32_t SMALL_STEP_MAX = 2048;     // unknowns of the step's system, padded (npp), and of the junction carried in
constexpr int SMALL_STEP_BLOCKS = 16;         // diagonal blocks of the kept factor's spine

// A chain step on a condensed block whose factor is kept (dnagpu_schur_carry_rhs), from dnagpu_block_load_reduced to the carried
// junction's renewed right-hand side: everything the separate kernels do, in their order of operations per element.
struct ChainRhsStep {
    // the condensed block (n_stn stations): its reduced right-hand side, where its stations' coordinates come from, its device block's vectors
    const double* red_rhs;
    const double* x_orig_src;
    const uint32_t* keep_idx;
    uint32_t n_stn;
    double* rhs;
    double* x_est;
    // the junction carried in (J = nullptr: none): S (ld npj, both triangles), the estimates S and r were formed at, r, its stations in this block
    const double* J;
    uint32_t npj;
    const double* jest_in;
    const double* jrhs_in;
    const uint32_t* idx_in;
    uint32_t k_in;
    // the step's kept factor in its light form (sym_inverse.h: sym_spine_async): X (ld npp), unknown order, spine blocks (elements)
    const double* X;
    const int32_t* map;
    uint32_t npp, nip, nj;
    int nblocks;
    uint32_t blk_o[SMALL_STEP_BLOCKS], blk_h[SMALL_STEP_BLOCKS];
    // the junction carried out: reduced right-hand side, linearisation point, its stations in this block
    double* jrhs_out;
    double* jest_out;
    const uint32_t* idx_out;
    uint32_t k_out;
};
void launch_chain_rhs_step(const ChainRhsStep& a, hipStream_t s);

// The two per-block steps of an iteration >= 2 of a GNSS-only network (a.reuse_factors) for MANY small blocks in ONE launch, a workgroup
// per block: small_condense = right-hand side from the measurements + its reduction with the kept factor (dna_adjust::CondenseBlock's reuse
// branch: dnagpu_form_rhs + dnagpu_partial_reduce_rhs); small_solve = the block's rigorous solve (PhasedForwardBlock / ...ReverseBlock /
// ...CombineBlock + UpdateEstimates* + UpdateEstimatesFinal, ADJ:2812-3057, 3512-3800): estimates back to the originals, right-hand side,
// the carried junctions' r + S dx, blocked substitution with the completed factor, estimates += corrections, largest correction,
// rigorous = estimated (and original = rigorous but for a last block).  A dnasegment-default cut has hundreds of such blocks: as separate
// kernels each of these steps was 10 - 25 launches and two host waits per block.
struct SmallBlockDesc {
    // right-hand side from the measurements (cluster_wb_kernel + form_rhs_kernel)
    const double* wblk;
    const uint32_t *vec_wrow, *vec_c0, *vec_k;
    uint32_t n_vec;
    const uint32_t *inc_off, *inc;
    const double* b;
    double *wb, *rhs, *corr, *corr_keep;
    double *x_orig, *x_est, *x_rig;
    uint32_t n_stn;
    // the block's kept factor, light form, completed (X_KK in its trailing block)
    const double* X;
    const int32_t* map;
    uint32_t npp, nip, nj;
    int nblocks;
    uint32_t blk_o[SMALL_STEP_BLOCKS], blk_h[SMALL_STEP_BLOCKS];
    double* red_rhs;
    // the junctions whose r + S (their estimates - ours) the rigorous solve adds, in this order (J = nullptr: none); information form
    const double* J[2];
    const double* jest[2];
    const double* jrhs[2];
    const uint32_t* jidx[2];
    uint32_t jk[2], jnp[2];
    uint32_t last;          // a last block of its network: forward solve (estimates as they are, originals untouched, corrections set aside)
    double* result;         // [0] the correction of largest magnitude (signed), [1] its row
};
void launch_small_condense(const SmallBlockDesc* table, uint32_t n, hipStream_t s);
void launch_small_solve(const SmallBlockDesc* table, uint32_t n, hipStream_t s);

}  // namespace dnagpu
