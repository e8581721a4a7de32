// Whole steps of the adjustment on small systems as one launch of one workgroup (small_steps.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnagpu {

constexpr uint32_t SMALL_STEP_MAX = 2048;     // unknowns of the step's system, padded (npp), and of the junction carried in
constexpr int SMALL_STEP_BLOCKS = 16;         // diagonal blocks of the kept factor's spine
constexpr size_t SMALL_STEP_BYTES = 2u << 20; // a right-hand-side-only chain step goes out as one launch while its factor is at most this large

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

// ---- chain plans: chain steps of the condensed schedule as DATA, taken through the elimination in lock-step batches ----------------
// A chain step (PhasedForwardBlock / PhasedReverseBlock on a condensed block, ADJ:2812 + 998-1281; a merge of two condensed systems; a step
// over a run's system) adds a few systems into one -- reduced blocks, junction matrices in information form, constraints --, eliminates
// all but the stations it carries on and leaves their complement.  A long chain of SMALL steps (a dnasegment-default cut: 666 blocks) is
// bound by its length, not its flops: with the steps described on the device the chain is cut into runs whose steps advance together, one
// batched launch sequence per step of ALL runs (dnagpu_chain_plan_*; dna_adjust::LockstepChains).
constexpr int CB_SRC_MAX = 3;
struct CbSrc {
    const double* F;        // the system added (ld np): lower triangle valid
    uint32_t np;
    const double* rhs;      // its right-hand side, by its own unknowns
    const double* jest;     // a junction matrix in information form: the estimates F and rhs were formed at; nullptr: a reduced system
    const uint32_t* pos;    // its station a is station pos[a] of the step's system
    const int32_t* inv;     // station of the step's system -> station of this source, -1: not in it
    uint32_t k;
};
struct CbStep {
    uint32_t n_stn, nj, k_out, nip, njp, npp, n_src;
    const double* const* est;    // per station: where its three coordinates are (the linearisation point); nullptr: none needed
    CbSrc src[CB_SRC_MAX];
    const double* con;      // 9 doubles per station: the constraints' 3 x 3 weights (column-major), nullptr: none
    const int32_t* map;     // the elimination's unknown order (npp): unknown of the step's system, -1 padding, -2 the right-hand side's row
    const uint32_t* keep;   // the k_out stations carried on, in the order of the output's unknowns
    double* X;              // the step's kept factor, light form (sym_inverse.h: sym_spine_async), npp x npp
    double* xe;             // scratch: 3 n_stn estimates ...
    double* rhs;            // ... and the assembled right-hand side
    double* outS;           // the complement (ld outnp, identity padded, both triangles)
    uint32_t outnp;
    double* out_rhs;        // its right-hand side
    double* out_jest;       // the kept stations' estimates (an information-form junction); nullptr: a reduced system
    int nblocks;
    uint32_t blk_o[SMALL_STEP_BLOCKS], blk_h[SMALL_STEP_BLOCKS];
};
struct CbMembers {
    double* F[32];          // the matrices the members of a batch are factored in (the chain's workspaces)
    double* X[32];          // ... and where their factors go: the plan's kept factors, or scratch of the chain when the plan keeps none
};
// members = table[0 .. nb): the right-hand sides and linearisation points (one workgroup per member) ...
void launch_cb_rhs(const CbStep* table, uint32_t nb, hipStream_t s);
// ... the systems, in elimination order, into the members' matrices (all members share the padded orders npp = nip + njp) ...
// (off / ext: the diagonal block of the members' matrices the systems go to -- 0 / npp for a chain step)
void launch_cb_assemble(const CbStep* table, uint32_t nb, const CbMembers& m, uint32_t npp, uint32_t off, uint32_t ext, hipStream_t s);
// ... and, after the elimination, complement / right-hand side / estimates out, the passenger row of the kept factor cleared
void launch_cb_post(const CbStep* table, uint32_t nb, const CbMembers& m, uint32_t nip, uint32_t npp, uint32_t outnp_max, hipStream_t s);
// the same steps with their factors kept (a.reuse_factors, iterations >= 2): right-hand sides only, any number of independent steps, one launch
void launch_cb_rhs_steps(const CbStep* table, uint32_t n, hipStream_t s);

}  // namespace dnagpu
