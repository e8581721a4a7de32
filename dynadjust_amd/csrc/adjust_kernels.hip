// HBM-bound kernels of the adjustment path: measurement weights, meas-minus-computed,
// deterministic normal-equation formation, sparse A^T W b, estimate update and the
// junction gather / scatter of the phased chain.
// Reference (under /root/reference/dynadjust/dynadjust/dnaadjust/dnaadjust.cpp):
//   LoadVarianceMatrix_G :4214, UpdateDesignMeasMatrices_GX :5283, UpdateNormals_G :1664,
//   Solve :6659 (AtVinv * b), UpdateEstimates* :3022, CarryStnEstimatesandVariances* :998/:1133/:3196.
//
// All sums are evaluated in a fixed order (CML order per matrix element), never with
// floating-point atomics, so that a run is bit-reproducible and the formation matches
// the CPU oracle bit for bit (contraction is off: the oracle is built with
// -ffp-contract=off too).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "adjust_kernels.h"
#include "terrestrial.h"

#pragma clang fp contract(off)

namespace dnagpu {

// index of symmetric 3x3 element (i,j) in the (xx, xy, yy, xz, yz, zz) storage
__device__ __forceinline__ int sym6(int i, int j) {
    int lo = i < j ? i : j, hi = i < j ? j : i;
    return hi * (hi + 1) / 2 + lo;
}

// W = V^-1 through the Cholesky factor V = U^T U (the reference calls
// dpotrf('U') + dpotri('U') on the 3x3, dnaadjust.cpp:4288 -> :8472).
__global__ void weights_kernel(const double* __restrict__ vcv6, const uint32_t* __restrict__ dst, double* __restrict__ wblk, uint32_t n,
                               int* __restrict__ bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* v = vcv6 + (size_t)i * 6;
    double v11 = v[0], v12 = v[1], v22 = v[2], v13 = v[3], v23 = v[4], v33 = v[5];
    double u11 = sqrt(v11);
    double u12 = v12 / u11;
    double u13 = v13 / u11;
    double d22 = v22 - u12 * u12;
    double u22 = sqrt(d22);
    double u23 = (v23 - u12 * u13) / u22;
    double d33 = (v33 - u13 * u13) - u23 * u23;
    double u33 = sqrt(d33);
    if (!(v11 > 0.0) || !(d22 > 0.0) || !(d33 > 0.0)) atomicMin(bad, (int)i);
    double t11 = 1.0 / u11, t22 = 1.0 / u22, t33 = 1.0 / u33;
    double t12 = -(t11 * u12) * t22;
    double t23 = -(t22 * u23) * t33;
    double t13 = -(t11 * (u12 * t23 + u13 * t33));
    double w0 = (t11 * t11 + t12 * t12) + t13 * t13;
    double w1 = t12 * t22 + t13 * t23;
    double w2 = t22 * t22 + t23 * t23;
    double w3 = t13 * t33;
    double w4 = t23 * t33;
    double w5 = t33 * t33;
    double* w = wblk + (size_t)dst[i] * 9;   // symmetric 3x3, column-major
    w[0] = w0; w[1] = w1; w[2] = w3;
    w[3] = w1; w[4] = w2; w[5] = w4;
    w[6] = w3; w[7] = w4; w[8] = w5;
}

// scatter the inverse of a k-vector cluster's 3k x 3k variance matrix into its k*k 3x3 blocks:
// block (j, j') element (ei, ej) = F(3j + ei, 3j' + ej)
__global__ void cluster_blocks_kernel(const double* __restrict__ F, uint32_t np, uint32_t k, double* __restrict__ wblk) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * k * 9) return;
    uint32_t blk = t / 9, e = t - blk * 9;
    uint32_t j = blk / k, jp = blk - j * k;
    uint32_t ei = e % 3, ej = e / 3;
    wblk[(size_t)blk * 9 + e] = F[(size_t)(3 * jp + ej) * np + 3 * j + ei];
}

// (xx, xy, yy, xz, yz, zz) of every vector's own weight block (diagnostics / tests)
__global__ void diag_weights_kernel(const double* __restrict__ wblk, const uint32_t* __restrict__ vec_wrow, const uint32_t* __restrict__ vec_c0,
                                    double* __restrict__ w6, uint32_t n_vec) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vec) return;
    const double* w = wblk + (size_t)(vec_wrow[v] + (v - vec_c0[v])) * 9;
    double* o = w6 + (size_t)v * 6;
    o[0] = w[0]; o[1] = w[3]; o[2] = w[4]; o[3] = w[6]; o[4] = w[7]; o[5] = w[8];
}

__global__ void compute_b_kernel(const uint32_t* __restrict__ s1, const uint32_t* __restrict__ s2, const double* __restrict__ obs,
                                 const double* __restrict__ xe, double* __restrict__ b, uint32_t n_bl) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_bl * 3) return;
    uint32_t i = t / 3, c = t - i * 3;
    // baseline: computed = x2 - x1 (ADJ:5304); point cluster (no first station): computed = x (ADJ:6343)
    double comp = xe[3 * s2[i] + c];
    if (s1[i] != 0xffffffffu) comp = comp - xe[3 * s1[i] + c];
    b[t] = obs[t] - comp;
}

// ResetAdjustment for one block in one launch: the coordinates `src` (device) go to the original, estimated (every chain) and rigorous
// coordinates, and measured-minus-computed of every chain is formed from them (the arithmetic of compute_b_kernel)
#define RESET_MAX_CHAINS 8
struct ResetBlockArgs {
    const double* src;
    double* x[2 + RESET_MAX_CHAINS];           // original, rigorous, estimated of chains 0 ..
    double* b[RESET_MAX_CHAINS];
    int nx, nb;                 // destinations in x / b
};
__global__ void reset_block_kernel(ResetBlockArgs a, const uint32_t* __restrict__ s1, const uint32_t* __restrict__ s2, const double* __restrict__ obs,
                                   uint32_t n3, uint32_t n_bl) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n3) {
        const double v = a.src[t];
        for (int q = 0; q < a.nx; ++q) a.x[q][t] = v;
    }
    if (t < n_bl * 3 && a.nb) {
        uint32_t i = t / 3, c = t - i * 3;
        double comp = a.src[3 * s2[i] + c];
        if (s1[i] != 0xffffffffu) comp = comp - a.src[3 * s1[i] + c];
        const double r = obs[t] - comp;
        for (int q = 0; q < a.nb; ++q) a.b[q][t] = r;
    }
}

// one thread per (station-pair block, element): N(3r+ei, 3c+ej) = sum over the pair's contributions
// (CML order) of +-W_block(ei, ej); a contribution = (index of a 3x3 weight block) << 1 | negative
__global__ void form_normals_kernel(const uint32_t* __restrict__ prow, const uint32_t* __restrict__ pcol,
                                    const uint32_t* __restrict__ poff, const uint32_t* __restrict__ pent,
                                    const double* __restrict__ wblk, double* __restrict__ F, uint32_t np, uint32_t n_pairs,
                                    uint32_t n_gnss_blk, uint32_t terr_shift) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs * 9) return;
    uint32_t p = t / 9, e = t - p * 9;
    int ei = e % 3, ej = e / 3;
    uint32_t r = prow[p], c = pcol[p];
    if (r == c && ei < ej) return;
    double s = 0.0;
    uint32_t k0 = poff[p], k1 = poff[p + 1];
    for (uint32_t k = k0; k < k1; ++k) {
        uint32_t ent = pent[k];
        // blocks of terrestrial measurements change with the estimates and exist once per chain: they follow the
        // (constant) GNSS weight blocks, chain c's copy terr_shift = c * n_terrestrial_blocks further on
        uint32_t blk = ent >> 1;
        if (blk >= n_gnss_blk) blk += terr_shift;
        double w = wblk[(size_t)blk * 9 + e];
        s += (ent & 1u) ? -w : w;
    }
    F[(size_t)(3 * c + ej) * np + 3 * r + ei] = s;
}

__global__ void add_diag3x3_kernel(double* __restrict__ F, uint32_t np, const uint32_t* __restrict__ stn,
                                   const double* __restrict__ w9, uint32_t k, double sign) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * 9) return;
    uint32_t q = t / 9, e = t - q * 9;
    int ei = e % 3, ej = e / 3;
    if (ei < ej) return;
    uint32_t s = stn[q];
    F[(size_t)(3 * s + ej) * np + 3 * s + ei] += sign * w9[(size_t)q * 9 + ej * 3 + ei];
}

// ---- the normals formed directly in the unknown order of the condensing step's elimination (dnagpu_block_form_reduce): what
// form_normals_kernel + add_diag3x3_kernel + schur_permute_kernel give, without the pass over the matrix in between.  spos[s] = position of
// station s's first unknown in that order (a station's three unknowns stay together and in order); lower triangle of F (ld).
__global__ __launch_bounds__(256) void init_ordered_kernel(double* __restrict__ F, uint32_t ld, const int32_t* __restrict__ map) {
    const uint32_t tr = blockIdx.x, tc = blockIdx.y;
    if (tc > tr) return;
    const uint32_t i = tr * 128 + (threadIdx.x & 127);
    const bool pad = map[i] == -1;
    for (uint32_t jl = threadIdx.x >> 7; jl < 128; jl += 2) {
        const uint32_t j = tc * 128 + jl;
        F[(size_t)j * ld + i] = (i == j && pad) ? 1.0 : 0.0;
    }
}
__global__ void form_normals_ordered_kernel(const uint32_t* __restrict__ prow, const uint32_t* __restrict__ pcol, const uint32_t* __restrict__ poff,
                                            const uint32_t* __restrict__ pent, const double* __restrict__ wblk, double* __restrict__ F, uint32_t ld,
                                            const uint32_t* __restrict__ spos, uint32_t n_pairs, uint32_t n_gnss_blk, uint32_t terr_shift) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs * 9) return;
    uint32_t p = t / 9, e = t - p * 9;
    int ei = e % 3, ej = e / 3;
    uint32_t r = prow[p], c = pcol[p];
    if (r == c && ei < ej) return;
    double s = 0.0;
    uint32_t k0 = poff[p], k1 = poff[p + 1];
    for (uint32_t k = k0; k < k1; ++k) {       // (same terms in the same order as form_normals_kernel: same bits)
        uint32_t ent = pent[k];
        uint32_t blk = ent >> 1;
        if (blk >= n_gnss_blk) blk += terr_shift;
        double w = wblk[(size_t)blk * 9 + e];
        s += (ent & 1u) ? -w : w;
    }
    const uint32_t pr = spos[r] + ei, pc = spos[c] + ej;
    if (pr >= pc)
        F[(size_t)pc * ld + pr] = s;
    else
        F[(size_t)pr * ld + pc] = s;
}
__global__ void add_diag3x3_ordered_kernel(double* __restrict__ F, uint32_t ld, const uint32_t* __restrict__ spos, const uint32_t* __restrict__ stn,
                                           const double* __restrict__ w9, uint32_t k, double sign) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k * 9) return;
    uint32_t q = t / 9, e = t - q * 9;
    int ei = e % 3, ej = e / 3;
    if (ei < ej) return;
    const uint32_t s0 = spos[stn[q]];
    F[(size_t)(s0 + ej) * ld + s0 + ei] += sign * w9[(size_t)q * 9 + ej * 3 + ei];
}
// the right-hand side as the passenger row `row` of F (row > every unknown's position)
__global__ void rhs_row_kernel(double* __restrict__ F, uint32_t ld, uint32_t row, const int32_t* __restrict__ map, const double* __restrict__ rhs,
                               uint32_t npp) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npp) return;
    const int32_t m = map[p];
    if (m >= 0) F[(size_t)p * ld + row] = rhs[m];
}
void launch_form_ordered(double* F, uint32_t ld, uint32_t npp, const int32_t* map, const uint32_t* spos, const uint32_t* prow, const uint32_t* pcol,
                         const uint32_t* poff, const uint32_t* pent, const double* wblk, uint32_t n_pairs, uint32_t n_gnss_blk, uint32_t terr_shift,
                         const uint32_t* con_stn, const double* con_w9, uint32_t n_con, const double* rhs, uint32_t rhs_row, hipStream_t s) {
    hipLaunchKernelGGL(init_ordered_kernel, dim3(npp / 128, npp / 128), dim3(256), 0, s, F, ld, map);
    if (n_pairs)
        hipLaunchKernelGGL(form_normals_ordered_kernel, dim3((n_pairs * 9 + 255) / 256), dim3(256), 0, s, prow, pcol, poff, pent, wblk, F, ld, spos,
                           n_pairs, n_gnss_blk, terr_shift);
    if (n_con)
        hipLaunchKernelGGL(add_diag3x3_ordered_kernel, dim3((n_con * 9 + 255) / 256), dim3(256), 0, s, F, ld, spos, con_stn, con_w9, n_con, 1.0);
    hipLaunchKernelGGL(rhs_row_kernel, dim3((npp + 255) / 256), dim3(256), 0, s, F, ld, rhs_row, map, rhs, npp);
}

// ---- the same for the members of a batch (dnagpu_block_form_reduce_batched): one launch per kernel for all of them, blockIdx.z = member.
// A dnasegment-default cut condenses 666 blocks in 42 batches: per member the kernels above (and the two of the right-hand side, the map's
// copy, the passenger row's memset, the extraction) were eleven launches of 3 - 8 us, 7 000 per iteration.  Same operations per element.
__global__ __launch_bounds__(256) void init_ordered_batch_kernel(const FormBatch fb, uint32_t ld) {
    const FormMember& m = fb.m[blockIdx.z];
    const uint32_t tr = blockIdx.x, tc = blockIdx.y;
    if (tc > tr) return;
    const uint32_t i = tr * 128 + (threadIdx.x & 127);
    const int32_t mi = m.map[i];
    if (tc == 0 && threadIdx.x < 128) m.map_out[i] = mi;         // (the kept factor's own copy of the order)
    const bool pad = mi == -1;
    for (uint32_t jl = threadIdx.x >> 7; jl < 128; jl += 2) {
        const uint32_t j = tc * 128 + jl;
        m.F[(size_t)j * ld + i] = (i == j && pad) ? 1.0 : 0.0;
    }
}
__global__ void form_normals_ordered_batch_kernel(const FormBatch fb, uint32_t ld) {
    const FormMember& m = fb.m[blockIdx.z];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.n_pairs * 9) return;
    uint32_t p = t / 9, e = t - p * 9;
    int ei = e % 3, ej = e / 3;
    uint32_t r = m.prow[p], c = m.pcol[p];
    if (r == c && ei < ej) return;
    double s = 0.0;
    uint32_t k0 = m.poff[p], k1 = m.poff[p + 1];
    for (uint32_t k = k0; k < k1; ++k) {
        uint32_t ent = m.pent[k];
        uint32_t blk = ent >> 1;
        if (blk >= m.n_gnss_blk) blk += m.terr_shift;
        double w = m.wblk[(size_t)blk * 9 + e];
        s += (ent & 1u) ? -w : w;
    }
    const uint32_t pr = m.spos[r] + ei, pc = m.spos[c] + ej;
    if (pr >= pc)
        m.F[(size_t)pc * ld + pr] = s;
    else
        m.F[(size_t)pr * ld + pc] = s;
}
__global__ void add_diag3x3_ordered_batch_kernel(const FormBatch fb, uint32_t ld) {
    const FormMember& m = fb.m[blockIdx.z];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.n_con * 9) return;
    uint32_t q = t / 9, e = t - q * 9;
    int ei = e % 3, ej = e / 3;
    if (ei < ej) return;
    const uint32_t s0 = m.spos[m.con_stn[q]];
    m.F[(size_t)(s0 + ej) * ld + s0 + ei] += m.con_w9[(size_t)q * 9 + ej * 3 + ei];
}
__global__ void rhs_row_batch_kernel(const FormBatch fb, uint32_t ld, uint32_t npp) {
    const FormMember& m = fb.m[blockIdx.z];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npp) return;
    const int32_t mp = m.map[p];
    if (mp >= 0) m.F[(size_t)p * ld + m.rhs_row] = m.rhs[mp];
}
void launch_form_ordered_batch(const FormBatch& fb, int nb, uint32_t ld, uint32_t npp, hipStream_t s) {
    uint32_t max_pairs = 0, max_con = 0;
    for (int b = 0; b < nb; ++b) {
        max_pairs = max_pairs > fb.m[b].n_pairs ? max_pairs : fb.m[b].n_pairs;
        max_con = max_con > fb.m[b].n_con ? max_con : fb.m[b].n_con;
    }
    hipLaunchKernelGGL(init_ordered_batch_kernel, dim3(npp / 128, npp / 128, nb), dim3(256), 0, s, fb, ld);
    if (max_pairs) hipLaunchKernelGGL(form_normals_ordered_batch_kernel, dim3((max_pairs * 9 + 255) / 256, 1, nb), dim3(256), 0, s, fb, ld);
    if (max_con) hipLaunchKernelGGL(add_diag3x3_ordered_batch_kernel, dim3((max_con * 9 + 255) / 256, 1, nb), dim3(256), 0, s, fb, ld);
    hipLaunchKernelGGL(rhs_row_batch_kernel, dim3((npp + 255) / 256, 1, nb), dim3(256), 0, s, fb, ld, npp);
}
// after the members' elimination: complement (both triangles, identity padded) and reduced right-hand side out of the trailing block of
// each member's matrix (schur_extract_kernel), the passenger row's entries in the kept factor's panels cleared
__global__ __launch_bounds__(256) void extract_batch_kernel(const ExtractBatch eb, uint32_t nip, uint32_t npp, uint32_t npj_max) {
    const ExtractMember& m = eb.m[blockIdx.z];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (j < npj_max) {
        if (j >= m.npj || i >= m.npj) return;
        double v = i == j ? 1.0 : 0.0;
        if (i < m.nj && j < m.nj) v = i >= j ? m.T[(size_t)j * npp + i] : m.T[(size_t)i * npp + j];
        m.S[(size_t)j * m.npj + i] = v;
        if (i == 0) m.r[j] = j < m.nj ? m.T[(size_t)j * npp + m.nj] : 0.0;
        return;
    }
    if (i < nip) m.X[(size_t)i * npp + nip + m.nj] = 0.0;
}
void launch_extract_batch(const ExtractBatch& eb, int nb, uint32_t nip, uint32_t npp, uint32_t npj_max, hipStream_t s) {
    const uint32_t rows = npj_max > nip ? npj_max : nip;
    hipLaunchKernelGGL(extract_batch_kernel, dim3((rows + 255) / 256, npj_max + 1, nb), dim3(256), 0, s, eb, nip, npp, npj_max);
}
// A^T W b of the members (cluster_wb_kernel + form_rhs_kernel)
__global__ void cluster_wb_batch_kernel(const RhsBatch rb) {
    const RhsMember& m = rb.m[blockIdx.z];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.n_vec * 3) return;
    uint32_t v = t / 3;
    int c = t - v * 3;
    const uint32_t k = m.vec_k[v], c0 = m.vec_c0[v];
    const double* row = m.wblk + (size_t)m.vec_wrow[v] * 9;
    double acc = 0.0;
    for (uint32_t jp = 0; jp < k; ++jp) {
        const double* w = row + (size_t)jp * 9;
        const double* bb = m.b + (size_t)(c0 + jp) * 3;
        double term = (w[c] * bb[0] + w[c + 3] * bb[1]) + w[c + 6] * bb[2];
        acc = (jp == 0) ? term : acc + term;
    }
    m.wb[t] = acc;
}
__global__ void form_rhs_batch_kernel(const RhsBatch rb) {
    const RhsMember& m = rb.m[blockIdx.z];
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.n_stn * 3) return;
    uint32_t s = t / 3;
    int c = t - s * 3;
    double acc = 0.0;
    for (uint32_t k = m.ioff[s]; k < m.ioff[s + 1]; ++k) {
        uint32_t e = m.inc[k];
        double v = m.wb[(size_t)(e >> 1) * 3 + c];
        acc += (e & 1u) ? v : -v;
    }
    m.rhs[t] = acc;
}
void launch_form_rhs_batch(const RhsBatch& rb, int nb, hipStream_t s) {
    uint32_t max_vec = 0, max_stn = 0;
    for (int b = 0; b < nb; ++b) {
        max_vec = max_vec > rb.m[b].n_vec ? max_vec : rb.m[b].n_vec;
        max_stn = max_stn > rb.m[b].n_stn ? max_stn : rb.m[b].n_stn;
    }
    if (max_vec) hipLaunchKernelGGL(cluster_wb_batch_kernel, dim3((max_vec * 3 + 255) / 256, 1, nb), dim3(256), 0, s, rb);
    if (max_stn) hipLaunchKernelGGL(form_rhs_batch_kernel, dim3((max_stn * 3 + 255) / 256, 1, nb), dim3(256), 0, s, rb);
}

// wb(v) = sum_j' W(v, j') b(j') over the vectors j' of v's cluster (the AtVinv columns of the cluster times b)
__global__ void cluster_wb_kernel(const double* __restrict__ wblk, const uint32_t* __restrict__ vec_wrow, const uint32_t* __restrict__ vec_c0,
                                  const uint32_t* __restrict__ vec_k, const double* __restrict__ b, double* __restrict__ wb, uint32_t n_vec) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_vec * 3) return;
    uint32_t v = t / 3;
    int c = t - v * 3;
    const uint32_t k = vec_k[v], c0 = vec_c0[v];
    const double* row = wblk + (size_t)vec_wrow[v] * 9;
    double acc = 0.0;
    for (uint32_t jp = 0; jp < k; ++jp) {
        const double* w = row + (size_t)jp * 9;
        const double* bb = b + (size_t)(c0 + jp) * 3;
        double term = (w[c] * bb[0] + w[c + 3] * bb[1]) + w[c + 6] * bb[2];
        acc = (jp == 0) ? term : acc + term;
    }
    wb[t] = acc;
}

// Post-adjustment statistics per GNSS vector (ComputePrecisionAdjMsrs_GX/_Y ADJ:8009/8037 + the chi-square terms of
// ComputeChiSquare_G/_XY ADJ:8530/8551).  One thread per vector:
//   prec6 = upper triangle (xx xy xz yy yz zz) of A S A^T with S = rigorous variances: (S22 - S12) - (S21 - S11) for a
//           baseline (Precision_Adjusted_GNSS_bsl, dnatemplatematrixfuncs.hpp:255), S22 for a point;
//   chi   = b . (W b) restricted to the vector's three rows (W b from cluster_wb_kernel).
// S is read from its lower triangle only, so the result does not depend on how the upper triangle was filled.
__global__ void msr_stats_kernel(const uint32_t* __restrict__ s1, const uint32_t* __restrict__ s2, const double* __restrict__ b,
                                 const double* __restrict__ wb, const double* __restrict__ S, uint32_t nps, double* __restrict__ prec6,
                                 double* __restrict__ chi, uint32_t n_vec) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vec) return;
    const double* bb = b + (size_t)v * 3;
    const double* ww = wb + (size_t)v * 3;
    chi[v] = (bb[0] * ww[0] + bb[1] * ww[1]) + bb[2] * ww[2];
    if (!S) return;
    auto sym = [&](uint32_t r, uint32_t c) { return r >= c ? S[(size_t)c * nps + r] : S[(size_t)r * nps + c]; };
    const uint32_t a = s1[v], e = 3 * s2[v];
    int q = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j, ++q) {
            double p = sym(e + i, e + j);
            if (a != 0xffffffffu) {
                const uint32_t f = 3 * a;
                p = (p - sym(f + i, e + j)) - (sym(e + i, f + j) - sym(f + i, f + j));
            }
            prec6[(size_t)v * 6 + q] = p;
        }
}

// ---- terrestrial measurements (one design row each; terrestrial.h) -----------------------------------------------------
// station records -> geodetic coordinates of the current estimates (UpdateGeographicCoords, dnaadjust.cpp:8711/8734)
__global__ void geodetic_kernel(const double* __restrict__ xe, double* __restrict__ llh, uint32_t n_stn) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_stn) return;
    dnagpu::tm::cart_to_geo(xe + 3 * (size_t)s, &llh[3 * (size_t)s], &llh[3 * (size_t)s + 1], &llh[3 * (size_t)s + 2]);
}

__device__ __forceinline__ dnagpu::tm::StationGeo load_geo(const double* llh, const double* geoid, const double* defl, uint32_t s) {
    dnagpu::tm::StationGeo g;
    g.lat = llh[3 * (size_t)s];
    g.lon = llh[3 * (size_t)s + 1];
    g.h = llh[3 * (size_t)s + 2];
    g.geoid = geoid[s];
    g.defl_v = defl[2 * (size_t)s];
    g.defl_m = defl[2 * (size_t)s + 1];
    return g;
}

// One thread per terrestrial measurement (FillDesignNormalMeasurementsMatrices for these types): computed value,
// meas-minus-computed, design row, then everything the formation kernels consume -- the 3x3 blocks w a_p^T a_q of its
// station pairs (p, q with local(p) >= local(q), the enumeration the host used for the pair lists) and the vectors
// a_p w b per station ("virtual" W b vectors behind the GNSS ones).
__global__ void tmsr_eval_kernel(const uint8_t* __restrict__ type, const uint32_t* __restrict__ stn, const double* __restrict__ val,
                                 const double* __restrict__ pre, const double* __restrict__ var, const double* __restrict__ ih,
                                 const double* __restrict__ th, const uint32_t* __restrict__ blk0, const uint32_t* __restrict__ vec0,
                                 const double* __restrict__ xe, const double* __restrict__ llh, const double* __restrict__ geoid,
                                 const double* __restrict__ defl, double* __restrict__ tb, double* __restrict__ trow,
                                 double* __restrict__ tblk, double* __restrict__ wb, uint32_t n_bl, uint32_t n_t) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_t) return;
    const char ty = (char)type[t];
    const int ns = dnagpu::tm::station_count(ty);
    uint32_t l[3] = {stn[3 * (size_t)t], stn[3 * (size_t)t + 1], stn[3 * (size_t)t + 2]};
    if (ns < 2) l[1] = l[0];
    if (ns < 3) l[2] = l[0];
    const double* X1 = xe + 3 * (size_t)l[0];
    const double* X2 = xe + 3 * (size_t)l[1];
    const double* X3 = xe + 3 * (size_t)l[2];
    const dnagpu::tm::StationGeo g1 = load_geo(llh, geoid, defl, l[0]), g2 = load_geo(llh, geoid, defl, l[1]);
    double row[9];
    const double value = dnagpu::tm::working_value(ty, val[t], pre[t], X1, X2, g1, g2);
    const double comp = dnagpu::tm::evaluate(ty, X1, X2, X3, g1, g2, ih[t], th[t], row);
    const double b = dnagpu::tm::meas_minus_comp(ty, value, comp);
    tb[t] = b;
#pragma unroll
    for (int i = 0; i < 9; ++i) trow[9 * (size_t)t + i] = row[i];
    if (ty == 'D') return;     // one angle of a direction set: blocks and vectors come from the set's dense weights (dset kernels)
    const double w = 1.0 / var[t];
    const double wbv = w * b;
    for (int q = 0; q < ns; ++q)
        for (int r = 0; r < 3; ++r) wb[3 * ((size_t)n_bl + vec0[t] + q) + r] = row[3 * q + r] * wbv;
    uint32_t idx = blk0[t];
    for (int p = 0; p < ns; ++p)
        for (int q = 0; q < ns; ++q) {
            if (p != q && !(l[p] > l[q])) continue;
            double* o = tblk + 9 * (size_t)idx++;
            for (int e = 0; e < 9; ++e) o[e] = (w * row[3 * p + e % 3]) * row[3 * q + e / 3];
        }
}

// Direction sets: N += A^T W A and rhs += A^T W b with the set's dense W (UpdateAtVinv_D / UpdateNormals_D, dnaadjust.cpp:1328, 1540).
// One thread per block (rows a, b of a set, station slots p, q): W_ab a_{a,p}^T a_{b,q}; then one thread per row: a_{a,p}^T sum_b W_ab b_b
__global__ void dset_blocks_kernel(const uint32_t* __restrict__ ra, const uint32_t* __restrict__ rb, const uint32_t* __restrict__ pq,
                                   const uint32_t* __restrict__ wi, const double* __restrict__ wts, const double* __restrict__ trow,
                                   double* __restrict__ out, uint32_t n) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double w = wts[wi[e]];
    const double* a = trow + 9 * (size_t)ra[e] + 3 * (pq[e] & 3u);
    const double* b = trow + 9 * (size_t)rb[e] + 3 * (pq[e] >> 2);
    double* o = out + 9 * (size_t)e;
    for (int i = 0; i < 9; ++i) o[i] = (w * a[i % 3]) * b[i / 3];
}
__global__ void dset_vectors_kernel(const uint32_t* __restrict__ row0, const uint32_t* __restrict__ kk, const uint32_t* __restrict__ woff,
                                    const double* __restrict__ wts, const double* __restrict__ tb, const double* __restrict__ trow,
                                    const uint32_t* __restrict__ vec0, double* __restrict__ wb, uint32_t n_bl, uint32_t n_t) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_t || kk[t] == 0) return;
    const uint32_t k = kk[t], a = t - row0[t];
    double s = 0.0;
    for (uint32_t b = 0; b < k; ++b) s += wts[woff[t] + a + (size_t)b * k] * tb[row0[t] + b];
    for (int q = 0; q < 3; ++q)
        for (int r = 0; r < 3; ++r) wb[3 * ((size_t)n_bl + vec0[t] + q) + r] = trow[9 * (size_t)t + 3 * q + r] * s;
}
void launch_dsets(const uint32_t* ra, const uint32_t* rb, const uint32_t* pq, const uint32_t* wi, const double* wts, const double* trow,
                  double* out, uint32_t n_blocks, const uint32_t* row0, const uint32_t* kk, const uint32_t* woff, const double* tb,
                  const uint32_t* vec0, double* wb, uint32_t n_bl, uint32_t n_t, hipStream_t s) {
    if (n_blocks) hipLaunchKernelGGL(dset_blocks_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, s, ra, rb, pq, wi, wts, trow, out, n_blocks);
    hipLaunchKernelGGL(dset_vectors_kernel, dim3((n_t + 255) / 256), dim3(256), 0, s, row0, kk, woff, wts, tb, trow, vec0, wb, n_bl, n_t);
}

// precision of the adjusted terrestrial measurements: a S a^T (ComputePrecisionAdjMsrs_A / _BCEKLMSVZ / _HIJPQR,
// dnaadjust.cpp:7877-8007); S read from its lower triangle
__global__ void tmsr_stats_kernel(const uint8_t* __restrict__ type, const uint32_t* __restrict__ stn, const double* __restrict__ trow,
                                  const double* __restrict__ S, uint32_t nps, double* __restrict__ prec, uint32_t n_t) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_t) return;
    const int ns = dnagpu::tm::station_count((char)type[t]);
    auto sym = [&](uint32_t r, uint32_t c) { return r >= c ? S[(size_t)c * nps + r] : S[(size_t)r * nps + c]; };
    const double* dr = trow + 9 * (size_t)t;
    double acc = 0.0;
    for (int s_ = 0; s_ < ns; ++s_)
        for (int i = 0; i < 3; ++i) {
            double part = 0.0;
            for (int j = 0; j < ns; ++j)
                for (int e = 0; e < 3; ++e) part += dr[3 * j + e] * sym(3 * stn[3 * (size_t)t + j] + e, 3 * stn[3 * (size_t)t + s_] + i);
            acc += part * dr[3 * s_ + i];
        }
    prec[t] = acc;
}

// rhs(3s+c) = sum over incident vectors (CML order) of +-(W b)_c
__global__ void form_rhs_kernel(const uint32_t* __restrict__ ioff, const uint32_t* __restrict__ inc, const double* __restrict__ wb,
                                double* __restrict__ rhs, uint32_t n_stn) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_stn * 3) return;
    uint32_t s = t / 3;
    int c = t - s * 3;
    double acc = 0.0;
    for (uint32_t k = ioff[s]; k < ioff[s + 1]; ++k) {
        uint32_t e = inc[k];
        double v = wb[(size_t)(e >> 1) * 3 + c];
        acc += (e & 1u) ? v : -v;
    }
    rhs[t] = acc;
}

// xe += corr ; (value, row) of the correction with the largest magnitude, first
// occurrence wins (matrix_2d::compute_maximum_value, dnamatrix_contiguous.cpp:1532)
__global__ __launch_bounds__(1024) void update_estimates_kernel(double* __restrict__ xe, const double* __restrict__ corr, uint32_t n,
                                                                double* __restrict__ out_val, uint32_t* __restrict__ out_idx) {
    __shared__ double sv[1024];
    __shared__ uint32_t si[1024];
    double best = -1.0;
    uint32_t bi = 0xffffffffu;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        double c = corr[i];
        xe[i] += c;
        double a = fabs(c);
        if (a > best) { best = a; bi = i; }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            double o = sv[threadIdx.x + s];
            uint32_t oi = si[threadIdx.x + s];
            if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = o;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t idx = si[0];
        *out_idx = idx;
        *out_val = (idx < n) ? corr[idx] : 0.0;
    }
}

// J(3a+ei, 3b+ej) = S(3 idx[a]+ei, 3 idx[b]+ej); S is full symmetric (N^-1)
__global__ void junction_gather_kernel(const double* __restrict__ S, uint32_t nps, const uint32_t* __restrict__ idx, uint32_t k,
                                       double* __restrict__ J, uint32_t npj) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // row of J
    uint32_t j = blockIdx.y;                             // column of J
    uint32_t nj = 3 * k;
    if (i >= nj || j >= nj) return;
    uint32_t si = 3 * idx[i / 3] + i % 3;
    uint32_t sj = 3 * idx[j / 3] + j % 3;
    J[(size_t)j * npj + i] = S[(size_t)sj * nps + si];
}

__global__ void gather_vec3_kernel(const double* __restrict__ x, const uint32_t* __restrict__ idx, uint32_t k, double* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * k) return;
    out[t] = x[3 * idx[t / 3] + t % 3];
}

// y[3 idx[a] + e] += v[3 a + e]
__global__ void scatter_add_vec3_kernel(double* __restrict__ y, const uint32_t* __restrict__ idx, uint32_t k, const double* __restrict__ v) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * k) return;
    y[3 * idx[t / 3] + t % 3] += v[t];
}
// y[3 dst[a] + e] = x[3 src[a] + e]
__global__ void copy_vec3_indexed_kernel(double* __restrict__ y, const uint32_t* __restrict__ dst, const double* __restrict__ x,
                                         const uint32_t* __restrict__ src, uint32_t k) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * k) return;
    y[3 * dst[t / 3] + t % 3] = x[3 * src[t / 3] + t % 3];
}

// D(3 idx[a]+ei, 3 idx[b]+ej) += J(3a+ei, 3b+ej) on the lower triangle of D
__global__ void junction_scatter_kernel(double* __restrict__ D, uint32_t npd, const uint32_t* __restrict__ idx, uint32_t k,
                                        const double* __restrict__ J, uint32_t npj) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t j = blockIdx.y;
    uint32_t nj = 3 * k;
    if (i >= nj || j >= nj) return;
    uint32_t di = 3 * idx[i / 3] + i % 3;
    uint32_t dj = 3 * idx[j / 3] + j % 3;
    if (di < dj) return;
    D[(size_t)dj * npd + di] += J[(size_t)j * npj + i];
}

// rhs[3 idx[a]+ei] += (jr ? jr[i] : 0) + sum_j J(i, j) * (jest[j] - xe[3 idx[j/3] + j%3])
// 16 rows per workgroup, the columns dealt round-robin to 16 lanes of threads, four accumulators per thread, the partial sums combined
// through LDS in a fixed order: deterministic (the first version ran one thread per row: 310 us for a 317-station junction)
__global__ __launch_bounds__(256) void junction_rhs_kernel(double* __restrict__ rhs, const double* __restrict__ xe, const uint32_t* __restrict__ idx,
                                                           uint32_t k, const double* __restrict__ J, uint32_t npj, const double* __restrict__ jest,
                                                           const double* __restrict__ jr) {
    // round 4: 16 rows x 16 column lanes per workgroup (it was 64 rows x 4): four times the workgroups and a quarter of the terms per thread --
    // on the path of every chain step this kernel is pure latency (35 us for a 150-station junction; ~10 us now)
    __shared__ double part[16][17];
    const uint32_t nj = 3 * k;
    const uint32_t r = threadIdx.x & 15, c = threadIdx.x >> 4;
    const uint32_t i = blockIdx.x * 16 + r;
    const uint32_t ic = i < nj ? i : nj - 1;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t j = c;
    for (; j + 48 < nj; j += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t jj = j + 16 * u;
            const double bj = jest[jj] - xe[3 * idx[jj / 3] + jj % 3];
            a[u] += J[(size_t)jj * npj + ic] * bj;
        }
    }
    for (; j < nj; j += 16) a[0] += J[(size_t)j * npj + ic] * (jest[j] - xe[3 * idx[j / 3] + j % 3]);
    part[c][r] = (a[0] + a[1]) + (a[2] + a[3]);
    __syncthreads();
    if (c == 0 && i < nj) {
        double s4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s4[q] = (part[4 * q][r] + part[4 * q + 1][r]) + (part[4 * q + 2][r] + part[4 * q + 3][r]);
        // jr (information form): the junction's own reduced right-hand side; the sum is then J * (the estimates it was formed at - xe),
        // zero when both blocks hold the same estimates of their common stations
        const double sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        rhs[3 * idx[i / 3] + i % 3] += jr ? jr[i] + sum : sum;
    }
}

// ---- launchers ---------------------------------------------------------------
void launch_weights(const double* vcv6, const uint32_t* dst, double* wblk, uint32_t n, int* bad, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(weights_kernel, dim3((n + 255) / 256), dim3(256), 0, s, vcv6, dst, wblk, n, bad);
}
void launch_cluster_blocks(const double* F, uint32_t np, uint32_t k, double* wblk, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(cluster_blocks_kernel, dim3((k * k * 9 + 255) / 256), dim3(256), 0, s, F, np, k, wblk);
}
void launch_diag_weights(const double* wblk, const uint32_t* vec_wrow, const uint32_t* vec_c0, double* w6, uint32_t n_vec, hipStream_t s) {
    if (!n_vec) return;
    hipLaunchKernelGGL(diag_weights_kernel, dim3((n_vec + 255) / 256), dim3(256), 0, s, wblk, vec_wrow, vec_c0, w6, n_vec);
}
void launch_compute_b(const uint32_t* s1, const uint32_t* s2, const double* obs, const double* xe, double* b, uint32_t n_bl, hipStream_t s) {
    if (!n_bl) return;
    hipLaunchKernelGGL(compute_b_kernel, dim3((n_bl * 3 + 255) / 256), dim3(256), 0, s, s1, s2, obs, xe, b, n_bl);
}
__global__ void block_table_kernel(const BlockTableRow* __restrict__ rows, int mode, int chains) {
    const BlockTableRow& r = rows[blockIdx.y];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const double* src = mode == 0 ? r.init : r.x_rig;
    if (t < r.n3) {
        const double v = src[t];
        if (mode == 0) {
            r.x_orig[t] = v;
            r.x_rig[t] = v;
        } else if (r.last) {
            r.x_orig[t] = v;
        }
        for (int c = 0; c < chains; ++c) r.x_est[c][t] = v;
    }
    if (t < r.nb3) {
        const uint32_t i = t / 3, c3 = t - i * 3;
        double comp = src[3 * r.s2[i] + c3];                        // (compute_b_kernel's arithmetic)
        if (r.s1[i] != 0xffffffffu) comp = comp - src[3 * r.s1[i] + c3];
        const double v = r.obs[t] - comp;
        for (int c = 0; c < chains; ++c) r.b[c][t] = v;
    }
}
void launch_block_table(const BlockTableRow* rows, uint32_t n, uint32_t max_len, int mode, int chains, hipStream_t s) {
    if (!n || !max_len) return;
    hipLaunchKernelGGL(block_table_kernel, dim3((max_len + 255) / 256, n), dim3(256), 0, s, rows, mode, chains);
}
void launch_reset_block(const double* src, double* x_orig, double* x_rig, double* const* x_est, double* const* b, int chains, bool with_b,
                        const uint32_t* s1, const uint32_t* s2, const double* obs, uint32_t n_stn, uint32_t n_bl, hipStream_t s) {
    ResetBlockArgs a{};
    if (chains > RESET_MAX_CHAINS) chains = RESET_MAX_CHAINS;      // (static_assert in dnagpu_api.hip: DNAGPU_NUM_CHAINS fits)
    a.src = src;
    a.x[0] = x_orig;
    a.x[1] = x_rig;
    for (int c = 0; c < chains; ++c) a.x[2 + c] = x_est[c];
    a.nx = 2 + chains;
    a.nb = (with_b && n_bl) ? chains : 0;
    for (int c = 0; c < a.nb; ++c) a.b[c] = b[c];
    const uint32_t n = std::max(3 * n_stn, a.nb ? 3 * n_bl : 0u);
    if (!n) return;
    hipLaunchKernelGGL(reset_block_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, s1, s2, obs, 3 * n_stn, n_bl);
}
void launch_form_normals(const uint32_t* prow, const uint32_t* pcol, const uint32_t* poff, const uint32_t* pent, const double* wblk, double* F,
                         uint32_t np, uint32_t n_pairs, uint32_t n_gnss_blk, uint32_t terr_shift, hipStream_t s) {
    if (!n_pairs) return;
    hipLaunchKernelGGL(form_normals_kernel, dim3((n_pairs * 9 + 255) / 256), dim3(256), 0, s, prow, pcol, poff, pent, wblk, F, np, n_pairs,
                       n_gnss_blk, terr_shift);
}
void launch_geodetic(const double* xe, double* llh, uint32_t n_stn, hipStream_t s) {
    if (!n_stn) return;
    hipLaunchKernelGGL(geodetic_kernel, dim3((n_stn + 127) / 128), dim3(128), 0, s, xe, llh, n_stn);
}
void launch_tmsr_eval(const uint8_t* type, const uint32_t* stn, const double* val, const double* pre, const double* var, const double* ih,
                      const double* th, const uint32_t* blk0, const uint32_t* vec0, const double* xe, const double* llh, const double* geoid,
                      const double* defl, double* tb, double* trow, double* tblk, double* wb, uint32_t n_bl, uint32_t n_t, hipStream_t s) {
    if (!n_t) return;
    hipLaunchKernelGGL(tmsr_eval_kernel, dim3((n_t + 63) / 64), dim3(64), 0, s, type, stn, val, pre, var, ih, th, blk0, vec0, xe, llh, geoid,
                       defl, tb, trow, tblk, wb, n_bl, n_t);
}
void launch_tmsr_stats(const uint8_t* type, const uint32_t* stn, const double* trow, const double* S, uint32_t nps, double* prec, uint32_t n_t,
                       hipStream_t s) {
    if (!n_t) return;
    hipLaunchKernelGGL(tmsr_stats_kernel, dim3((n_t + 63) / 64), dim3(64), 0, s, type, stn, trow, S, nps, prec, n_t);
}
void launch_add_diag3x3(double* F, uint32_t np, const uint32_t* stn, const double* w9, uint32_t k, double sign, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(add_diag3x3_kernel, dim3((k * 9 + 255) / 256), dim3(256), 0, s, F, np, stn, w9, k, sign);
}
void launch_form_rhs(const double* wblk, const uint32_t* vec_wrow, const uint32_t* vec_c0, const uint32_t* vec_k, const double* b, double* wb,
                     uint32_t n_vec, const uint32_t* ioff, const uint32_t* inc, double* rhs, uint32_t n_stn, hipStream_t s) {
    if (n_vec) hipLaunchKernelGGL(cluster_wb_kernel, dim3((n_vec * 3 + 255) / 256), dim3(256), 0, s, wblk, vec_wrow, vec_c0, vec_k, b, wb, n_vec);
    if (!n_stn) return;
    hipLaunchKernelGGL(form_rhs_kernel, dim3((n_stn * 3 + 255) / 256), dim3(256), 0, s, ioff, inc, wb, rhs, n_stn);
}
void launch_msr_stats(const double* wblk, const uint32_t* vec_wrow, const uint32_t* vec_c0, const uint32_t* vec_k, const uint32_t* s1,
                      const uint32_t* s2, const double* b, double* wb, const double* S, uint32_t nps, double* prec6, double* chi, uint32_t n_vec,
                      hipStream_t s) {
    if (!n_vec) return;
    hipLaunchKernelGGL(cluster_wb_kernel, dim3((n_vec * 3 + 255) / 256), dim3(256), 0, s, wblk, vec_wrow, vec_c0, vec_k, b, wb, n_vec);
    hipLaunchKernelGGL(msr_stats_kernel, dim3((n_vec + 255) / 256), dim3(256), 0, s, s1, s2, b, wb, S, nps, prec6, chi, n_vec);
}
// dna_adjust::UpdateIterationDiagnostics (ADJ:7450-7554), one block's visit: every station's correction of this iteration against the
// correction the station was last seen with (the previous iteration -- or, for a station shared with an earlier block, that block's
// visit in THIS iteration: the reference walks the blocks in order with one record per station).  Anti-parallel (cos < -0.5), similar in
// size (ratio 0.3 ... 3), not both below a millimetre: the station's cycle count goes up, otherwise back to zero.  visit[s] = the count
// where it has reached 2 (the host then keeps the history record), 0 elsewhere; *flagged counts those.  One thread per station; the
// stations of a block are distinct, the blocks follow each other on one stream.
__global__ void osc_update_kernel(const double* __restrict__ corr, const uint32_t* __restrict__ gidx, uint32_t n_stn, double* __restrict__ prev,
                                  uint32_t* __restrict__ seen, uint32_t* __restrict__ cnt, uint32_t* __restrict__ visit, uint32_t* __restrict__ flagged) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_stn) return;
    const uint32_t g = gidx[s];
    const double cx = corr[3 * s], cy = corr[3 * s + 1], cz = corr[3 * s + 2];
    visit[s] = 0;
    if (!seen[g]) {
        seen[g] = 1;
        prev[3 * (size_t)g] = cx;
        prev[3 * (size_t)g + 1] = cy;
        prev[3 * (size_t)g + 2] = cz;
        return;
    }
    const double px = prev[3 * (size_t)g], py = prev[3 * (size_t)g + 1], pz = prev[3 * (size_t)g + 2];
    prev[3 * (size_t)g] = cx;
    prev[3 * (size_t)g + 1] = cy;
    prev[3 * (size_t)g + 2] = cz;
    const double magCurr = sqrt(cx * cx + cy * cy + cz * cz), magPrev = sqrt(px * px + py * py + pz * pz);
    if (magCurr < 0.001 && magPrev < 0.001) {
        cnt[g] = 0;
        return;
    }
    const double dot = cx * px + cy * py + cz * pz, denom = magCurr * magPrev;
    const double cosAngle = denom > 1e-30 ? dot / denom : 0.0;
    const double ratio = magPrev > 1e-30 ? magCurr / magPrev : 0.0;
    uint32_t c = cnt[g];
    c = (cosAngle < -0.5 && ratio > 0.3 && ratio < 3.0) ? c + 1 : 0;
    cnt[g] = c;
    if (c >= 2) {
        visit[s] = c;
        atomicAdd(flagged, 1u);
    }
}
// The same for MANY blocks in one launch (a dnasegment-default cut: 666 launches of ~3 us per iteration otherwise).  A station shared by
// several blocks is visited by each of them IN ORDER -- the state it is compared with is the previous visit's, possibly a block earlier in
// this very iteration -- so the launch goes by STATION: a thread takes one station of the network through its visits (block, position in
// the block) in block order, its state in registers.  off / visits: the stations' visit lists (CSR), built once per set of blocks.
__global__ __launch_bounds__(256) void osc_update_stations_kernel(const OscRow* __restrict__ rows, const uint32_t* __restrict__ off,
                                                                  const uint2* __restrict__ visits, uint32_t n_global, double* __restrict__ prev,
                                                                  uint32_t* __restrict__ seen, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flagged) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_global) return;
    const uint32_t v0 = off[g], v1 = off[g + 1];
    if (v0 == v1) return;
    bool was_seen = seen[g] != 0;
    double px = prev[3 * (size_t)g], py = prev[3 * (size_t)g + 1], pz = prev[3 * (size_t)g + 2];
    uint32_t c = cnt[g], hits = 0;
    for (uint32_t v = v0; v < v1; ++v) {
        const uint2 w = visits[v];
        const OscRow r = rows[w.x];
        const uint32_t s = w.y;
        const double cx = r.corr[3 * s], cy = r.corr[3 * s + 1], cz = r.corr[3 * s + 2];
        uint32_t mark = 0;
        if (was_seen) {
            const double magCurr = sqrt(cx * cx + cy * cy + cz * cz), magPrev = sqrt(px * px + py * py + pz * pz);
            if (magCurr < 0.001 && magPrev < 0.001) {
                c = 0;
            } else {
                const double dot = cx * px + cy * py + cz * pz, denom = magCurr * magPrev;
                const double cosAngle = denom > 1e-30 ? dot / denom : 0.0;
                const double ratio = magPrev > 1e-30 ? magCurr / magPrev : 0.0;
                c = (cosAngle < -0.5 && ratio > 0.3 && ratio < 3.0) ? c + 1 : 0;
                if (c >= 2) {
                    mark = c;
                    ++hits;
                }
            }
        }
        r.visit[s] = mark;
        was_seen = true;
        px = cx;
        py = cy;
        pz = cz;
    }
    seen[g] = 1;
    prev[3 * (size_t)g] = px;
    prev[3 * (size_t)g + 1] = py;
    prev[3 * (size_t)g + 2] = pz;
    cnt[g] = c;
    if (hits) atomicAdd(flagged, hits);
}
void launch_osc_update_stations(const OscRow* rows, const uint32_t* off, const void* visits, uint32_t n_global, double* prev, uint32_t* seen, uint32_t* cnt,
                                uint32_t* flagged, hipStream_t s) {
    if (n_global)
        hipLaunchKernelGGL(osc_update_stations_kernel, dim3((n_global + 255) / 256), dim3(256), 0, s, rows, off, (const uint2*)visits, n_global, prev, seen, cnt, flagged);
}
void launch_osc_update(const double* corr, const uint32_t* gidx, uint32_t n_stn, double* prev, uint32_t* seen, uint32_t* cnt, uint32_t* visit,
                       uint32_t* flagged, hipStream_t s) {
    if (n_stn) hipLaunchKernelGGL(osc_update_kernel, dim3((n_stn + 255) / 256), dim3(256), 0, s, corr, gidx, n_stn, prev, seen, cnt, visit, flagged);
}

void launch_update_estimates(double* xe, const double* corr, uint32_t n, double* out_val, uint32_t* out_idx, hipStream_t s) {
    hipLaunchKernelGGL(update_estimates_kernel, dim3(1), dim3(1024), 0, s, xe, corr, n, out_val, out_idx);
}
// ---- junction carry by partial elimination (dnagpu_schur_carry) ----------------------------------------------------
// dst (ldd, npp x npp, lower tiles) = the normals with the unknowns re-ordered by `map`: map[i] >= 0 is the unknown of src
// (lower triangle valid, lds) that sits at row / column i; -1 marks identity padding; -2 the single row that carries the
// right-hand side, which the elimination then reduces like any other row (forward substitution for free).
// One workgroup per 128 x 128 tile of the lower tile triangle; the strict upper part of diagonal tiles is zeroed.
__global__ __launch_bounds__(256) void schur_permute_kernel(const double* __restrict__ src, uint32_t lds, const int32_t* __restrict__ map,
                                                            const double* __restrict__ rhs, double* __restrict__ dst, uint32_t ldd) {
    const uint32_t tr = blockIdx.x, tc = blockIdx.y;
    if (tc > tr) return;
    const uint32_t il = threadIdx.x & 127;
    const uint32_t i = tr * 128 + il;
    const int32_t mi = map[i];
    // (round 4: a tile's 128 columns over four workgroups -- blockIdx.z -- and the gathers of a thread unrolled: 43 -> ~12 us for the 7 x 7
    //  tiles of a condensed block, on the path of every chain step)
#pragma unroll 8
    for (uint32_t jl = blockIdx.z * 32 + (threadIdx.x >> 7); jl < blockIdx.z * 32 + 32; jl += 2) {
        const uint32_t j = tc * 128 + jl;
        const int32_t mj = map[j];
        double v = 0.0;
        if (i >= j) {
            if (mi >= 0 && mj >= 0) {
                const uint32_t r = mi > mj ? mi : mj, c = mi > mj ? mj : mi;
                v = src[(size_t)c * lds + r];
            } else if (mi == -2) {
                v = mj >= 0 ? rhs[mj] : 0.0;
            } else if (mi == -1 && i == j) {
                v = 1.0;
            }
        }
        dst[(size_t)j * ldd + i] = v;
    }
}

// T (ldt): trailing block after the elimination -- rows / columns 0..nj-1 the Schur complement (lower), row nj the reduced
// right-hand side.  S and S2 (npj x npj, identity padded, both triangles) receive the complement, r the reduced rhs.
__global__ void schur_extract_kernel(const double* __restrict__ T, uint32_t ldt, uint32_t nj, uint32_t npj, double* __restrict__ S,
                                     double* __restrict__ S2, double* __restrict__ r) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= npj) return;
    double v = i == j ? 1.0 : 0.0;
    if (i < nj && j < nj) v = i >= j ? T[(size_t)j * ldt + i] : T[(size_t)i * ldt + j];
    S[(size_t)j * npj + i] = v;
    if (S2) S2[(size_t)j * npj + i] = v;
    if (i == 0) r[j] = j < nj ? T[(size_t)j * ldt + nj] : 0.0;
}

// junction estimates = block estimates + corrections of the junction unknowns
__global__ void schur_estimates_kernel(const double* __restrict__ xe, const uint32_t* __restrict__ idx, uint32_t k,
                                       const double* __restrict__ delta, double* __restrict__ jest) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * k) return;
    jest[t] = xe[3 * idx[t / 3] + t % 3] + delta[t];
}

// dnagpu_partial_reduce_rhs: right-hand side in the elimination's order
__global__ void gather_map_kernel(const double* __restrict__ rhs, const int32_t* __restrict__ map, uint32_t npp, double* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npp) return;
    const int32_t m = map[p];
    out[p] = m >= 0 ? rhs[m] : 0.0;
}
// part[c][i] = sum over the columns j of chunk c (j <= i only when `lower`) of A(i, j) x(j), i < rows; deterministic like symv
// (round 5: a thread owns TWO adjacent rows -- 16-byte loads -- and keeps four columns in flight: the panel products of the substitution ran
//  at 2.6 TB/s with one 8-byte load and two accumulators per thread; A and lda are 16-byte aligned by construction: offsets of whole tiles)
__global__ __launch_bounds__(256) void gemv_partial_kernel(const double* __restrict__ A, uint32_t lda, uint32_t rows, uint32_t cols,
                                                           const double* __restrict__ x, double* __restrict__ part, uint32_t cols_per_chunk,
                                                           int lower) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const uint32_t i = 2 * (blockIdx.x * 256 + threadIdx.x);
    const uint32_t c = blockIdx.y;
    if (i >= rows) return;
    const bool pair = i + 1 < rows;
    uint32_t j0 = c * cols_per_chunk, j1 = j0 + cols_per_chunk;
    if (j1 > cols) j1 = cols;
    if (lower && j1 > i + 2) j1 = i + 2;          // (row i + 1 reaches column i + 1; row i's term there is masked below)
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    uint32_t j = j0;
    if (pair) {
        for (; j + 4 <= j1; j += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const d2 m = *reinterpret_cast<const d2*>(A + (size_t)(j + u) * lda + i);
                const double xv = x[j + u];
                a[u] += ((lower && j + u > i) ? 0.0 : m.x) * xv;
                b[u] += m.y * xv;
            }
        }
        for (; j < j1; ++j) {
            const d2 m = *reinterpret_cast<const d2*>(A + (size_t)j * lda + i);
            a[0] += ((lower && j > i) ? 0.0 : m.x) * x[j];
            b[0] += m.y * x[j];
        }
        part[(size_t)c * rows + i] = (a[0] + a[1]) + (a[2] + a[3]);
        part[(size_t)c * rows + i + 1] = (b[0] + b[1]) + (b[2] + b[3]);
    } else {
        if (lower && j1 > i + 1) j1 = i + 1;
        for (; j < j1; ++j) a[0] += A[(size_t)j * lda + i] * x[j];
        part[(size_t)c * rows + i] = a[0];
    }
}
// out[i] = (base ? base[i] : 0) + sign * sum_c part[c][i], i < n_out (part rows = rows)
__global__ void gemv_finish_kernel(const double* __restrict__ part, uint32_t rows, uint32_t nchunks, const double* __restrict__ base, double sign,
                                   double* __restrict__ out, uint32_t n_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    double s = 0.0;
    for (uint32_t c = 0; c < nchunks; ++c) s += part[(size_t)c * rows + i];
    out[i] = (base ? base[i] : 0.0) + sign * s;
}
// out[j] = sum over i >= j of A(i, j) y(i), j < n: the transposed product with a lower triangular matrix (column-major, so a wave
// walks ONE column: 64 lanes x 8 B contiguous).  One wave per column, lane partial sums in a fixed order, butterfly reduction:
// deterministic.  HBM-bound: n^2 / 2 x 8 B.
__global__ __launch_bounds__(256) void gemv_t_lower_kernel(const double* __restrict__ A, uint32_t lda, uint32_t n, const double* __restrict__ y,
                                                           double* __restrict__ out) {
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (j >= n) return;
    const double* col = A + (size_t)j * lda;
    // (round 5: 16-byte loads, two rows per lane, two such loads in flight: 1 KiB segments from an aligned start)
    typedef double d2 __attribute__((ext_vector_type(2)));
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    uint32_t i = (j & ~127u) + 2 * lane;
    for (; i + 128 + 1 < n; i += 256) {
        const d2 m0 = *reinterpret_cast<const d2*>(col + i), m1 = *reinterpret_cast<const d2*>(col + i + 128);
        a0 += (i >= j ? m0.x : 0.0) * y[i];
        a1 += (i + 1 >= j ? m0.y : 0.0) * y[i + 1];
        a2 += m1.x * y[i + 128];
        a3 += m1.y * y[i + 129];
    }
    for (; i < n; i += 128) {
        if (i >= j) a0 += col[i] * y[i];
        if (i + 1 < n && i + 1 >= j) a1 += col[i + 1] * y[i + 1];
    }
    double v = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) out[j] = v;
}
// out[j] = base[j] + sign * sum over i < rows of A(i, j) y(i), j < cols: the transposed product with a rectangular panel, same scheme
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ A, uint32_t lda, uint32_t rows, uint32_t cols,
                                                     const double* __restrict__ y, const double* __restrict__ base, double sign,
                                                     double* __restrict__ out) {
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (j >= cols) return;
    const double* col = A + (size_t)j * lda;
    typedef double d2 __attribute__((ext_vector_type(2)));
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    uint32_t i = 2 * lane;
    for (; i + 128 + 1 < rows; i += 256) {
        const d2 m0 = *reinterpret_cast<const d2*>(col + i), m1 = *reinterpret_cast<const d2*>(col + i + 128);
        a0 += m0.x * y[i];
        a1 += m0.y * y[i + 1];
        a2 += m1.x * y[i + 128];
        a3 += m1.y * y[i + 129];
    }
    for (; i < rows; i += 128) {
        a0 += col[i] * y[i];
        if (i + 1 < rows) a1 += col[i + 1] * y[i + 1];
    }
    double v = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) out[j] = (base ? base[j] : 0.0) + sign * v;
}
// out[map[p]] = v[p] for the positions that carry an unknown
__global__ void scatter_map_kernel(const double* __restrict__ v, const int32_t* __restrict__ map, uint32_t npp, double* __restrict__ out) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npp) return;
    const int32_t m = map[p];
    if (m >= 0) out[m] = v[p];
}
void launch_gemv_t_lower(const double* A, uint32_t lda, uint32_t n, const double* y, double* out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(gemv_t_lower_kernel, dim3((n + 3) / 4), dim3(256), 0, s, A, lda, n, y, out);
}
void launch_gemv_t(const double* A, uint32_t lda, uint32_t rows, uint32_t cols, const double* y, const double* base, double sign, double* out,
                   hipStream_t s) {
    if (!cols) return;
    hipLaunchKernelGGL(gemv_t_kernel, dim3((cols + 3) / 4), dim3(256), 0, s, A, lda, rows, cols, y, base, sign, out);
}
void launch_scatter_map(const double* v, const int32_t* map, uint32_t npp, double* out, hipStream_t s) {
    if (!npp) return;
    hipLaunchKernelGGL(scatter_map_kernel, dim3((npp + 255) / 256), dim3(256), 0, s, v, map, npp, out);
}
void launch_gather_map(const double* rhs, const int32_t* map, uint32_t npp, double* out, hipStream_t s) {
    hipLaunchKernelGGL(gather_map_kernel, dim3((npp + 255) / 256), dim3(256), 0, s, rhs, map, npp, out);
}
void launch_gemv(const double* A, uint32_t lda, uint32_t rows, uint32_t cols, const double* x, double* part, uint32_t nchunks, int lower,
                 const double* base, double sign, double* out, uint32_t n_out, hipStream_t s) {
    if (!n_out) return;      // (nothing to produce: a zero-sized grid is an invalid launch configuration)
    if (!rows || !cols) {
        hipLaunchKernelGGL(gemv_finish_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, part, rows, 0u, base, sign, out, n_out);
        return;
    }
    const uint32_t cpc = (cols + nchunks - 1) / nchunks;
    hipLaunchKernelGGL(gemv_partial_kernel, dim3((rows + 511) / 512, nchunks), dim3(256), 0, s, A, lda, rows, cols, x, part, cpc, lower);
    hipLaunchKernelGGL(gemv_finish_kernel, dim3((n_out + 255) / 256), dim3(256), 0, s, part, rows, nchunks, base, sign, out, n_out);
}

// dnagpu_partial_complete: the kept block (kk, npk x npk, both triangles) goes back behind the eliminated part: trailing
// njp x njp block of the retained matrix, identity beyond the nj kept unknowns (the row that carried the rhs included)
__global__ void partial_set_trailing_kernel(double* __restrict__ T, uint32_t ldt, uint32_t njp, const double* __restrict__ kk, uint32_t npk,
                                            uint32_t nj) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;
    if (i >= njp) return;
    T[(size_t)j * ldt + i] = (i < nj && j < nj) ? kk[(size_t)j * npk + i] : (i == j ? 1.0 : 0.0);
}

// inverse in the elimination's unknown order (F, ldf) -> natural order (inv, np; already identity padded)
__global__ __launch_bounds__(256) void unpermute_kernel(const double* __restrict__ F, uint32_t ldf, const int32_t* __restrict__ map,
                                                        double* __restrict__ inv, uint32_t np) {
    const uint32_t i = blockIdx.x * 128 + (threadIdx.x & 127);
    const int32_t mi = map[i];
    if (mi < 0) return;
    // (round 5: a tile's 128 columns over four workgroups -- blockIdx.z -- and a thread's 16 loads in flight together: a 768 x 768 matrix of a
    //  small block took 62 us on 36 workgroups of 64 dependent load / store pairs each)
    const uint32_t j0 = blockIdx.y * 128 + blockIdx.z * 32 + (threadIdx.x >> 7);
    double v[16];
    int32_t mj[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        mj[t] = map[j0 + 2 * t];
        v[t] = F[(size_t)(j0 + 2 * t) * ldf + i];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
        if (mj[t] >= 0) inv[(size_t)mj[t] * np + mi] = v[t];
}

// the members of a batch (dnagpu_partial_finish_batched): identity padding of every member's inverse (init_padding_kernel) and its un-permutation,
// two launches for all members
__global__ void init_padding_batch_kernel(const UnpermuteBatch ub) {
    const UnpermuteMember& m = ub.m[blockIdx.y];
    const uint32_t j = blockIdx.x;
    if (j >= m.np) return;
    for (uint32_t i = (j >= m.n ? 0 : m.n) + threadIdx.x; i < m.np; i += blockDim.x) m.inv[(size_t)j * m.np + i] = (i == j) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void unpermute_batch_kernel(const UnpermuteBatch ub, uint32_t ldf) {
    const UnpermuteMember& m = ub.m[blockIdx.z >> 2];
    const uint32_t i = blockIdx.x * 128 + (threadIdx.x & 127);
    const int32_t mi = m.map[i];
    if (mi < 0) return;
    const uint32_t j0 = blockIdx.y * 128 + (blockIdx.z & 3) * 32 + (threadIdx.x >> 7);
    double v[16];
    int32_t mj[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        mj[t] = m.map[j0 + 2 * t];
        v[t] = m.F[(size_t)(j0 + 2 * t) * ldf + i];
    }
#pragma unroll
    for (int t = 0; t < 16; ++t)
        if (mj[t] >= 0) m.inv[(size_t)mj[t] * m.np + mi] = v[t];
}
void launch_unpermute_batch(const UnpermuteBatch& ub, int nb, uint32_t npp, hipStream_t s) {
    uint32_t np_max = 0;
    bool pad = false;
    for (int b = 0; b < nb; ++b) {
        np_max = np_max > ub.m[b].np ? np_max : ub.m[b].np;
        pad = pad || ub.m[b].np > ub.m[b].n;
    }
    if (pad) hipLaunchKernelGGL(init_padding_batch_kernel, dim3(np_max, nb), dim3(128), 0, s, ub);
    hipLaunchKernelGGL(unpermute_batch_kernel, dim3(npp / 128, npp / 128, 4 * nb), dim3(256), 0, s, ub, npp);
}

void launch_partial_set_trailing(double* T, uint32_t ldt, uint32_t njp, const double* kk, uint32_t npk, uint32_t nj, hipStream_t s) {
    hipLaunchKernelGGL(partial_set_trailing_kernel, dim3((njp + 255) / 256, njp), dim3(256), 0, s, T, ldt, njp, kk, npk, nj);
}
void launch_unpermute(const double* F, uint32_t ldf, uint32_t npp, const int32_t* map, double* inv, uint32_t np, hipStream_t s) {
    hipLaunchKernelGGL(unpermute_kernel, dim3(npp / 128, npp / 128, 4), dim3(256), 0, s, F, ldf, map, inv, np);
}

void launch_schur_permute(const double* src, uint32_t lds, const int32_t* map, const double* rhs, double* dst, uint32_t ldd, uint32_t npp,
                          hipStream_t s) {
    hipLaunchKernelGGL(schur_permute_kernel, dim3(npp / 128, npp / 128, 4), dim3(256), 0, s, src, lds, map, rhs, dst, ldd);
}
void launch_schur_extract(const double* T, uint32_t ldt, uint32_t nj, uint32_t npj, double* S, double* S2, double* r, hipStream_t s) {
    hipLaunchKernelGGL(schur_extract_kernel, dim3((npj + 255) / 256, npj), dim3(256), 0, s, T, ldt, nj, npj, S, S2, r);
}
void launch_schur_estimates(const double* xe, const uint32_t* idx, uint32_t k, const double* delta, double* jest, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(schur_estimates_kernel, dim3((3 * k + 255) / 256), dim3(256), 0, s, xe, idx, k, delta, jest);
}

void launch_junction_gather(const double* S, uint32_t nps, const uint32_t* idx, uint32_t k, double* J, uint32_t npj, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(junction_gather_kernel, dim3((3 * k + 255) / 256, 3 * k), dim3(256), 0, s, S, nps, idx, k, J, npj);
}
void launch_gather_vec3(const double* x, const uint32_t* idx, uint32_t k, double* out, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(gather_vec3_kernel, dim3((3 * k + 255) / 256), dim3(256), 0, s, x, idx, k, out);
}
void launch_scatter_add_vec3(double* y, const uint32_t* idx, uint32_t k, const double* v, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(scatter_add_vec3_kernel, dim3((3 * k + 255) / 256), dim3(256), 0, s, y, idx, k, v);
}
void launch_copy_vec3_indexed(double* y, const uint32_t* dst, const double* x, const uint32_t* src, uint32_t k, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(copy_vec3_indexed_kernel, dim3((3 * k + 255) / 256), dim3(256), 0, s, y, dst, x, src, k);
}
void launch_junction_scatter(double* D, uint32_t npd, const uint32_t* idx, uint32_t k, const double* J, uint32_t npj, hipStream_t s) {
    if (!k) return;
    hipLaunchKernelGGL(junction_scatter_kernel, dim3((3 * k + 255) / 256, 3 * k), dim3(256), 0, s, D, npd, idx, k, J, npj);
}
void launch_junction_rhs(double* rhs, const double* xe, const uint32_t* idx, uint32_t k, const double* J, uint32_t npj, const double* jest,
                         const double* jr, hipStream_t s) {
    hipLaunchKernelGGL(junction_rhs_kernel, dim3((3 * k + 15) / 16), dim3(256), 0, s, rhs, xe, idx, k, J, npj, jest, jr);
}

}  // namespace dnagpu
