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

}  // namespace dnagpu
