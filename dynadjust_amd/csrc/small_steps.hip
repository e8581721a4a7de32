// Whole steps of the adjustment on SMALL systems as one launch of one workgroup.
//
// A dnasegment-default cut (150 stations per block, include/config/dnaoptions.hpp:382) gives hundreds of blocks whose condensed
// systems have a few hundred unknowns.  From the second iteration of a GNSS-only network on (a.reuse_factors) a chain step on such a
// system factors nothing: it assembles a right-hand side of a few hundred numbers and takes it through a kept factor of a megabyte --
// as separate kernels that is ten launches of 3 - 5 us each for ~2 us of work, 665 times in a row per direction (dnasegment150:
// 62 ms of chains per iteration).  Here the step is ONE kernel: the vectors live in LDS, the factor streams through once.
// Replaces, for iterations >= 2, what the reference does in every iteration per block: Solve() + CarryStnEstimatesandVariancesForward /
// ...Reverse (dnaadjust.cpp:2812, 998-1281), in the form of dnagpu_schur_carry_rhs (dnagpu_api.hip).
#include <hip/hip_runtime.h>
#include "small_steps.h"

namespace dnagpu {

namespace {

constexpr int NT = 1024;            // threads of the workgroup (16 waves)
constexpr int NW = NT / 64;
// The one-workgroup kernels below keep up to five vectors of SMALL_STEP_MAX doubles and the column slices' partial sums in static LDS
// (5 x 16 KiB + 8 KiB): more than the 64 KiB a workgroup has on gfx90a / gfx942.  They are written for the 160 KiB of gfx950 (the only
// target of this library's build, __graft_entry__.py) and take a CU to themselves.
static_assert((5 * SMALL_STEP_MAX + NW * 64) * sizeof(double) + 256 <= 160 * 1024, "the vectors of a small step must fit the LDS of one gfx950 workgroup");

// y[i] = base[i] + sign * sum_{j < cols, (LOWER: j <= i)} A[i + j * lda] * x[j],  i < rows.   x, base, y: LDS (y may be base, not x);
// part: NW * 64 doubles of LDS.  Rows in groups of 64 (a lane per row: coalesced columns), the waves left over split the columns
// (interleaved), eight independent accumulators per lane, the column slices added in a fixed order: deterministic.
template <bool LOWER>
__device__ void wg_matvec(const double* __restrict__ A, uint32_t lda, uint32_t rows, uint32_t cols, const double* x, const double* base, double sign,
                          double* y, double* part) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (rows == 0) return;                   // (uniform; nothing to write, and no row group to deal the waves to)
    const uint32_t RG = (rows + 63) / 64;
    if (RG >= (uint32_t)NW) {
        for (uint32_t rg = w; rg < RG; rg += NW) {
            const uint32_t i = rg * 64 + lane;
            const uint32_t ic = i < rows ? i : rows - 1;
            const uint32_t jend = LOWER ? min(cols, rg * 64 + 64) : cols;
            double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t j = 0;
            for (; j + 8 <= jend; j += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double m = A[(size_t)(j + u) * lda + ic];
                    a[u] += (LOWER && j + u > i) ? 0.0 : m * x[j + u];
                }
            }
            for (; j < jend; ++j) {
                const double m = A[(size_t)j * lda + ic];
                a[0] += (LOWER && j > i) ? 0.0 : m * x[j];
            }
            const double s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
            if (i < rows) y[i] = (base ? base[i] : 0.0) + sign * s;
        }
        __syncthreads();
        return;
    }
    const uint32_t CS = NW / RG;              // column slices per row group
    if ((uint32_t)w < RG * CS) {
        const uint32_t rg = w % RG, cs = w / RG;
        const uint32_t i = rg * 64 + lane;
        const uint32_t ic = i < rows ? i : rows - 1;
        const uint32_t jend = LOWER ? min(cols, rg * 64 + 64) : cols;
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t j = cs;
        for (; j + 7 * CS < jend; j += 8 * CS) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t jj = j + u * CS;
                const double m = A[(size_t)jj * lda + ic];
                a[u] += (LOWER && jj > i) ? 0.0 : m * x[jj];
            }
        }
        for (; j < jend; j += CS) {
            const double m = A[(size_t)j * lda + ic];
            a[0] += (LOWER && j > i) ? 0.0 : m * x[j];
        }
        part[(cs * RG + rg) * 64 + lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    for (uint32_t i = tid; i < rows; i += NT) {
        const uint32_t rg = i >> 6, l = i & 63;
        double s = 0.0;
        for (uint32_t cs = 0; cs < CS; ++cs) s += part[(cs * RG + rg) * 64 + l];
        y[i] = (base ? base[i] : 0.0) + sign * s;
    }
    __syncthreads();
}

// y[j] = base[j] + sign * sum_{i < rows, (LOWER: i >= j)} A[i + j * lda] * x[i],  j < cols: the transposed product.  A wave per column
// (64 lanes x 8 B contiguous), four accumulators per lane, butterfly reduction: deterministic.  x, base, y: LDS (y may be base, not x).
template <bool LOWER>
__device__ void wg_matvec_t(const double* __restrict__ A, uint32_t lda, uint32_t rows, uint32_t cols, const double* x, const double* base, double sign,
                            double* y) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (cols == 0) return;                   // (uniform)
    for (uint32_t j = w; j < cols; j += NW) {
        const double* col = A + (size_t)j * lda;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        uint32_t i = (LOWER ? (j & ~63u) : 0u) + lane;
        for (; i + 192 < rows; i += 256) {
            const double m0 = col[i], m1 = col[i + 64], m2 = col[i + 128], m3 = col[i + 192];
            a0 += ((LOWER && i < j) ? 0.0 : m0) * x[i];
            a1 += m1 * x[i + 64];
            a2 += m2 * x[i + 128];
            a3 += m3 * x[i + 192];
        }
        for (; i < rows; i += 64) a0 += ((LOWER && i < j) ? 0.0 : col[i]) * x[i];
        double v = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) y[j] = (base ? base[j] : 0.0) + sign * v;
    }
    __syncthreads();
}

// rhs (LDS, 3 n_stn) <- A^T W b of the block: wb(v) = sum_j' W(v, j') b(j') per vector (cluster_wb_kernel), then per station the signed sum
// over its incident vectors in CML order (form_rhs_kernel) -- the same operations in the same order as the separate kernels
__device__ void wg_form_rhs(const SmallBlockDesc& d, double* rhs) {
    const int tid = threadIdx.x;
    for (uint32_t t = tid; t < 3 * d.n_vec; t += NT) {
        const uint32_t v = t / 3;
        const int c = (int)(t - v * 3);
        const uint32_t k = d.vec_k[v], c0 = d.vec_c0[v];
        const double* row = d.wblk + (size_t)d.vec_wrow[v] * 9;
        double acc = 0.0;
        for (uint32_t jp = 0; jp < k; ++jp) {
            const double* w = row + (size_t)jp * 9;
            const double* bb = d.b + (size_t)(c0 + jp) * 3;
            const double term = (w[c] * bb[0] + w[c + 3] * bb[1]) + w[c + 6] * bb[2];
            acc = (jp == 0) ? term : acc + term;
        }
        d.wb[t] = acc;
    }
    __syncthreads();
    for (uint32_t t = tid; t < 3 * d.n_stn; t += NT) {
        const uint32_t s = t / 3;
        const int c = (int)(t - s * 3);
        double acc = 0.0;
        for (uint32_t k = d.inc_off[s]; k < d.inc_off[s + 1]; ++k) {
            const uint32_t e = d.inc[k];
            const double v = d.wb[(size_t)(e >> 1) * 3 + c];
            acc += (e & 1u) ? v : -v;
        }
        rhs[t] = acc;
    }
    __syncthreads();
}

// v (LDS, npp) <- rhs in the elimination's order, then the forward half of the blocked substitution (only what is not padding streams in)
__device__ void wg_forward(const SmallBlockDesc& d, const double* rhs, double* v, double* t, double* part) {
    const int tid = threadIdx.x;
    for (uint32_t p = tid; p < d.npp; p += NT) {
        const int32_t m = d.map[p];
        v[p] = m >= 0 ? rhs[m] : 0.0;
    }
    __syncthreads();
    const uint32_t ni = 3 * d.n_stn - d.nj, last = d.nip + d.nj;
    for (int q = 0; q < d.nblocks; ++q) {
        const uint32_t o = d.blk_o[q], h = d.blk_h[q];
        if (o >= ni) break;
        const uint32_t hr = min(h, ni - o), below = last - (o + h);
        wg_matvec<true>(d.X + (size_t)o * d.npp + o, d.npp, hr, hr, v + o, nullptr, 1.0, t, part);
        for (uint32_t i = tid; i < hr; i += NT) v[o + i] = t[i];
        __syncthreads();
        if (below) wg_matvec<false>(d.X + (size_t)o * d.npp + o + h, d.npp, below, hr, v + o, v + o + h, -1.0, v + o + h, part);
    }
}

}  // namespace

__global__ __launch_bounds__(NT) void small_condense_kernel(const SmallBlockDesc* __restrict__ table) {
    __shared__ double rhs[SMALL_STEP_MAX], v[SMALL_STEP_MAX], t[SMALL_STEP_MAX], part[NW * 64];
    const SmallBlockDesc& d = table[blockIdx.x];
    const int tid = threadIdx.x;
    wg_form_rhs(d, rhs);
    for (uint32_t i = tid; i < 3 * d.n_stn; i += NT) d.rhs[i] = rhs[i];
    wg_forward(d, rhs, v, t, part);
    for (uint32_t i = tid; i < d.nj; i += NT) d.red_rhs[i] = v[d.nip + i];
}

__global__ __launch_bounds__(NT) void small_solve_kernel(const SmallBlockDesc* __restrict__ table) {
    __shared__ double rhs[SMALL_STEP_MAX], v[SMALL_STEP_MAX], t[SMALL_STEP_MAX], u[SMALL_STEP_MAX], part[NW * 64];
    __shared__ int any_dx;
    __shared__ double best_v[NW];
    __shared__ uint32_t best_i[NW];
    const SmallBlockDesc& d = table[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t n = 3 * d.n_stn;
    // estimates back to the originals (PrepareAdjustmentReverse / ...Combine, ADJ:3157, 3863); a last block's forward solve starts from its estimates
    if (!d.last)
        for (uint32_t i = tid; i < n; i += NT) d.x_est[i] = d.x_orig[i];
    if (tid == 0) any_dx = 0;
    wg_form_rhs(d, rhs);            // (ends with a barrier: the estimates above are visible to the workgroup)
    // the carried junctions, information form: rhs[their stations] += r + S (the estimates S and r were formed at - ours)
    for (int q = 0; q < 2; ++q) {
        if (!d.J[q]) continue;
        const uint32_t nj = 3 * d.jk[q];
        int mine = 0;
        for (uint32_t j = tid; j < nj; j += NT) {
            const double dx = d.jest[q][j] - d.x_est[3 * d.jidx[q][j / 3] + j % 3];
            t[j] = dx;
            mine |= dx != 0.0;
        }
        if (mine) any_dx = 1;
        __syncthreads();
        const bool nonzero = any_dx != 0;
        __syncthreads();
        if (tid == 0) any_dx = 0;
        if (nonzero) {
            wg_matvec<false>(d.J[q], d.jnp[q], nj, nj, t, nullptr, 1.0, u, part);
        } else {
            for (uint32_t j = tid; j < nj; j += NT) u[j] = 0.0;
            __syncthreads();
        }
        for (uint32_t j = tid; j < nj; j += NT) rhs[3 * d.jidx[q][j / 3] + j % 3] += d.jrhs[q][j] + u[j];
        __syncthreads();
    }
    for (uint32_t i = tid; i < n; i += NT) d.rhs[i] = rhs[i];
    wg_forward(d, rhs, v, t, part);
    // the kept block: v_K <- X_KK^T (X_KK v_K)
    const uint32_t ni = n - d.nj, last = d.nip + d.nj;
    {
        const double* XK = d.X + (size_t)d.nip * d.npp + d.nip;
        wg_matvec<true>(XK, d.npp, d.nj, d.nj, v + d.nip, nullptr, 1.0, t, part);
        wg_matvec_t<true>(XK, d.npp, d.nj, d.nj, t, nullptr, 1.0, v + d.nip);
    }
    // backward: v_b <- X_bb^T (v_b - L_(below, b)^T v_below), from the last diagonal block of the eliminated part to the first
    for (int q = d.nblocks - 1; q >= 0; --q) {
        const uint32_t o = d.blk_o[q], h = d.blk_h[q];
        if (o >= ni) continue;
        const uint32_t hr = min(h, ni - o), below = last - (o + h);
        if (below) {
            wg_matvec_t<false>(d.X + (size_t)o * d.npp + o + h, d.npp, below, hr, v + o + h, v + o, -1.0, u);
        } else {
            for (uint32_t i = tid; i < hr; i += NT) u[i] = v[o + i];
            __syncthreads();
        }
        wg_matvec_t<true>(d.X + (size_t)o * d.npp + o, d.npp, hr, hr, u, nullptr, 1.0, v + o);
    }
    // corrections in the block's own order; estimates += corrections; the correction of largest magnitude, first occurrence
    // (matrix_2d::compute_maximum_value, MAT:1532); rigorous = estimated; original = rigorous (UpdateEstimatesFinal, ADJ:3744) but for a last block
    for (uint32_t p = tid; p < d.npp; p += NT) {
        const int32_t m = d.map[p];
        if (m >= 0) rhs[m] = v[p];
    }
    __syncthreads();
    double best = -1.0;
    uint32_t bi = 0xffffffffu;
    for (uint32_t i = tid; i < n; i += NT) {
        const double c = rhs[i];
        d.corr[i] = c;
        if (d.last) d.corr_keep[i] = c;
        const double x = d.x_est[i] + c;
        d.x_est[i] = x;
        d.x_rig[i] = x;
        if (!d.last) d.x_orig[i] = x;
        const double a = fabs(c);
        if (a > best) {
            best = a;
            bi = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const uint32_t oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) {
        best_v[w] = best;
        best_i[w] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < NW; ++q)
            if (best_v[q] > best || (best_v[q] == best && best_i[q] < bi)) {
                best = best_v[q];
                bi = best_i[q];
            }
        d.result[0] = bi < n ? rhs[bi] : 0.0;
        d.result[1] = (double)bi;
    }
}

void launch_small_condense(const SmallBlockDesc* table, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(small_condense_kernel, dim3(n), dim3(NT), 0, s, table);
}
void launch_small_solve(const SmallBlockDesc* table, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(small_solve_kernel, dim3(n), dim3(NT), 0, s, table);
}

__global__ __launch_bounds__(NT) void chain_rhs_step_kernel(const ChainRhsStep a) {
    __shared__ double rhs[SMALL_STEP_MAX], xe[SMALL_STEP_MAX], v[SMALL_STEP_MAX], t[SMALL_STEP_MAX], part[NW * 64];
    __shared__ int any_dx;
    const int tid = threadIdx.x;
    const uint32_t n = 3 * a.n_stn;
    // the step's system: rhs <- the condensed block's reduced right-hand side, linearisation point <- the source block's originals
    if (tid == 0) any_dx = 0;
    for (uint32_t i = tid; i < n; i += NT) {
        rhs[i] = a.red_rhs[i];
        const double x = a.x_orig_src[3 * a.keep_idx[i / 3] + i % 3];
        xe[i] = x;
        a.x_est[i] = x;
    }
    __syncthreads();
    // the junction carried in (information form): rhs[its stations] += r + S (the estimates S and r were formed at - ours)
    if (a.J) {
        const uint32_t nj = 3 * a.k_in;
        int mine = 0;
        for (uint32_t j = tid; j < nj; j += NT) {
            const double d = a.jest_in[j] - xe[3 * a.idx_in[j / 3] + j % 3];
            t[j] = d;
            mine |= d != 0.0;
        }
        if (mine) any_dx = 1;
        __syncthreads();
        if (any_dx) {                         // (zero whenever both blocks hold the same estimates of their common stations)
            wg_matvec<false>(a.J, a.npj, nj, nj, t, nullptr, 1.0, v, part);
        } else {
            for (uint32_t j = tid; j < nj; j += NT) v[j] = 0.0;
            __syncthreads();
        }
        for (uint32_t j = tid; j < nj; j += NT) rhs[3 * a.idx_in[j / 3] + j % 3] += a.jrhs_in[j] + v[j];
        __syncthreads();
    }
    for (uint32_t i = tid; i < n; i += NT) a.rhs[i] = rhs[i];
    // into the elimination's order, then the forward half of the blocked substitution with the kept factor
    for (uint32_t p = tid; p < a.npp; p += NT) {
        const int32_t m = a.map[p];
        v[p] = m >= 0 ? rhs[m] : 0.0;
    }
    __syncthreads();
    // (only what is not padding streams in: the eliminated unknowns up to ni -- beyond them the factor is an identity and the vector zero --
    //  and the rows down to the last kept unknown; one CU pulls ~25 GB/s, so the padded megabyte of a 150-unknown step was 40 of its 48 us)
    const uint32_t ni = n - a.nj, last = a.nip + a.nj;
    for (int q = 0; q < a.nblocks; ++q) {
        const uint32_t o = a.blk_o[q], h = a.blk_h[q];
        if (o >= ni) break;
        const uint32_t hr = min(h, ni - o), below = last - (o + h);
        wg_matvec<true>(a.X + (size_t)o * a.npp + o, a.npp, hr, hr, v + o, nullptr, 1.0, t, part);
        for (uint32_t i = tid; i < hr; i += NT) v[o + i] = t[i];
        __syncthreads();
        if (below) wg_matvec<false>(a.X + (size_t)o * a.npp + o + h, a.npp, below, hr, v + o, v + o + h, -1.0, v + o + h, part);
    }
    for (uint32_t i = tid; i < a.nj; i += NT) a.jrhs_out[i] = v[a.nip + i];
    for (uint32_t i = tid; i < 3 * a.k_out; i += NT) a.jest_out[i] = xe[3 * a.idx_out[i / 3] + i % 3];
}

void launch_chain_rhs_step(const ChainRhsStep& a, hipStream_t s) {
    hipLaunchKernelGGL(chain_rhs_step_kernel, dim3(1), dim3(NT), 0, s, a);
}


// ---- chain plans (small_steps.h) ------------------------------------------------------------------------------------------------------
namespace {

// rhs, xe (LDS, 3 n_stn) <- the step's right-hand side and linearisation point: every source's right-hand side at its stations; an
// information-form junction adds r + S (the estimates S and r were formed at - ours), as dnagpu_junction_rhs / chain_rhs_step_kernel do
__device__ void wg_step_rhs(const CbStep& d, double* rhs, double* xe, double* t, double* v, double* part, int* any_dx) {
    const int tid = threadIdx.x;
    const uint32_t n = 3 * d.n_stn;
    if (tid == 0) *any_dx = 0;
    for (uint32_t i = tid; i < n; i += NT) {
        rhs[i] = 0.0;
        xe[i] = d.est ? d.est[i / 3][i % 3] : 0.0;
    }
    __syncthreads();
    for (uint32_t q = 0; q < d.n_src; ++q) {
        const CbSrc& sc = d.src[q];
        const uint32_t nq = 3 * sc.k;
        if (sc.jest) {
            int mine = 0;
            for (uint32_t j = tid; j < nq; j += NT) {
                const double dx = sc.jest[j] - xe[3 * sc.pos[j / 3] + j % 3];
                t[j] = dx;
                mine |= dx != 0.0;
            }
            if (mine) *any_dx = 1;
            __syncthreads();
            const bool nonzero = *any_dx != 0;
            __syncthreads();
            if (tid == 0) *any_dx = 0;
            if (nonzero) {
                wg_matvec<false>(sc.F, sc.np, nq, nq, t, nullptr, 1.0, v, part);
            } else {
                for (uint32_t j = tid; j < nq; j += NT) v[j] = 0.0;
                __syncthreads();
            }
            for (uint32_t j = tid; j < nq; j += NT) rhs[3 * sc.pos[j / 3] + j % 3] += sc.rhs[j] + v[j];
        } else {
            for (uint32_t j = tid; j < nq; j += NT) rhs[3 * sc.pos[j / 3] + j % 3] += sc.rhs[j];
        }
        __syncthreads();
    }
}

}  // namespace

__global__ __launch_bounds__(NT) void cb_rhs_kernel(const CbStep* __restrict__ table) {
    __shared__ double rhs[SMALL_STEP_MAX], xe[SMALL_STEP_MAX], v[SMALL_STEP_MAX], t[SMALL_STEP_MAX], part[NW * 64];
    __shared__ int any_dx;
    const CbStep& d = table[blockIdx.x];
    wg_step_rhs(d, rhs, xe, t, v, part, &any_dx);
    for (uint32_t i = threadIdx.x; i < 3 * d.n_stn; i += NT) {
        d.rhs[i] = rhs[i];
        d.xe[i] = xe[i];
    }
}

// The step's system in the elimination's order (what dnagpu_block_load_reduced + add_diag3x3 + junction_scatter + schur_permute leave):
// element (i, j), i >= j, is the sum over the sources that hold both unknowns, plus the constraint block of the station; the row map = -2
// carries the right-hand side; identity on the padding's diagonal.  One workgroup per quarter of a 128 x 128 tile of the lower tile triangle.
// The sources hold BOTH triangles (schur_extract_kernel / cb_post_kernel write them so).
// (off: the block of the member's matrix -- rows and columns off .. off + ext - 1 -- the system goes to: 0 for a chain step; the trailing block of
//  a kept factor's matrix for a step that eliminates nothing, d.map = nullptr: the system in its own order, identity beyond it)
__global__ __launch_bounds__(256) void cb_assemble_kernel(const CbStep* __restrict__ table, const CbMembers mem, uint32_t npp, uint32_t off) {
    const uint32_t b = blockIdx.z >> 2, qz = blockIdx.z & 3;
    const CbStep& d = table[b];
    double* __restrict__ dst = mem.F[b] + (size_t)off * npp + off;
    const uint32_t n3 = 3 * d.n_stn;
    const uint32_t tr = blockIdx.x, tc = blockIdx.y;
    if (tc > tr) return;
    const uint32_t il = threadIdx.x & 127;
    const uint32_t i = tr * 128 + il;
    const int32_t mi = d.map ? d.map[i] : (i < n3 ? (int32_t)i : -1);
    const uint32_t su = mi >= 0 ? (uint32_t)mi / 3 : 0, eu = mi >= 0 ? (uint32_t)mi % 3 : 0;
    const uint32_t nsrc = d.n_src;
    int32_t ai[CB_SRC_MAX];
#pragma unroll
    for (int q = 0; q < CB_SRC_MAX; ++q) ai[q] = (mi >= 0 && (uint32_t)q < nsrc) ? d.src[q].inv[su] : -1;
    // a thread's 16 columns: their map entries, then their source stations, then the matrix entries -- each round of loads in flight together
    // (one column at a time the kernel was three dependent loads deep per element: 42 us for a batch of sixteen 640-unknown systems)
    const uint32_t j0 = tc * 128 + qz * 32 + (threadIdx.x >> 7);
    int32_t mj[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) mj[t] = d.map ? d.map[j0 + 2 * t] : (j0 + 2 * t < n3 ? (int32_t)(j0 + 2 * t) : -1);
    double v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = 0.0;
    if (mi >= 0) {
#pragma unroll
        for (int q = 0; q < CB_SRC_MAX; ++q) {
            if (ai[q] < 0) continue;         // (uniform over most of a wave: a source covers whole station ranges)
            const CbSrc& sc = d.src[q];
            int32_t bj[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) bj[t] = (mj[t] >= 0 && i >= j0 + 2 * t) ? sc.inv[(uint32_t)mj[t] / 3] : -1;
            const size_t r0 = 3 * (size_t)ai[q] + eu;
            double f[16];
            // (every source holds both triangles: no swap to the lower one, the rows of a column stay coalesced)
#pragma unroll
            for (int t = 0; t < 16; ++t) f[t] = bj[t] >= 0 ? sc.F[(size_t)(3 * (uint32_t)bj[t] + (uint32_t)mj[t] % 3) * sc.np + r0] : 0.0;
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] += f[t];
        }
        if (d.con) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (mj[t] >= 0 && i >= j0 + 2 * t && (uint32_t)mj[t] / 3 == su) v[t] += d.con[(size_t)su * 9 + ((uint32_t)mj[t] % 3) * 3 + eu];
        }
    } else if (mi == -2) {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = mj[t] >= 0 ? d.rhs[mj[t]] : 0.0;
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = (i == j0 + 2 * t) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const uint32_t j = j0 + 2 * t;
        dst[(size_t)j * npp + i] = i >= j ? v[t] : 0.0;
    }
}

// after the elimination: the trailing block of a member's matrix holds the complement (lower) and, in row nj, the reduced right-hand side
__global__ __launch_bounds__(256) void cb_post_kernel(const CbStep* __restrict__ table, const CbMembers mem, uint32_t nip, uint32_t npp,
                                                      uint32_t outnp_max) {
    const CbStep& d = table[blockIdx.z];
    const double* __restrict__ T = mem.F[blockIdx.z] + (size_t)nip * npp + nip;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (j < outnp_max) {
        if (j >= d.outnp || i >= d.outnp) return;
        double v = i == j ? 1.0 : 0.0;
        if (i < d.nj && j < d.nj) v = i >= j ? T[(size_t)j * npp + i] : T[(size_t)i * npp + j];
        d.outS[(size_t)j * d.outnp + i] = v;
        if (i == 0 && j < d.nj) d.out_rhs[j] = T[(size_t)j * npp + d.nj];
        return;
    }
    // (the row that carried the right-hand side is no unknown: its entries in the kept factor's panels go)
    if (i < nip) mem.X[blockIdx.z][(size_t)i * npp + nip + d.nj] = 0.0;
    if (d.out_jest && i < 3 * d.k_out) d.out_jest[i] = d.xe[3 * d.keep[i / 3] + i % 3];
}

// a right-hand-side-only step with its factor kept: chain_rhs_step_kernel from a table, for any number of independent steps
__global__ __launch_bounds__(NT) void cb_rhs_steps_kernel(const CbStep* __restrict__ table) {
    __shared__ double rhs[SMALL_STEP_MAX], xe[SMALL_STEP_MAX], v[SMALL_STEP_MAX], t[SMALL_STEP_MAX], part[NW * 64];
    __shared__ int any_dx;
    const CbStep& d = table[blockIdx.x];
    const int tid = threadIdx.x;
    const uint32_t n = 3 * d.n_stn;
    wg_step_rhs(d, rhs, xe, t, v, part, &any_dx);
    for (uint32_t p = tid; p < d.npp; p += NT) {
        const int32_t m = d.map[p];
        v[p] = m >= 0 ? rhs[m] : 0.0;
    }
    __syncthreads();
    const uint32_t ni = n - d.nj, last = d.nip + d.nj;
    for (int q = 0; q < d.nblocks; ++q) {
        const uint32_t o = d.blk_o[q], h = d.blk_h[q];
        if (o >= ni) break;
        const uint32_t hr = min(h, ni - o), below = last - (o + h);
        wg_matvec<true>(d.X + (size_t)o * d.npp + o, d.npp, hr, hr, v + o, nullptr, 1.0, t, part);
        for (uint32_t i = tid; i < hr; i += NT) v[o + i] = t[i];
        __syncthreads();
        if (below) wg_matvec<false>(d.X + (size_t)o * d.npp + o + h, d.npp, below, hr, v + o, v + o + h, -1.0, v + o + h, part);
    }
    for (uint32_t i = tid; i < d.nj; i += NT) d.out_rhs[i] = v[d.nip + i];
    if (d.out_jest)
        for (uint32_t i = tid; i < 3 * d.k_out; i += NT) d.out_jest[i] = xe[3 * d.keep[i / 3] + i % 3];
}

void launch_cb_rhs(const CbStep* table, uint32_t nb, hipStream_t s) {
    if (nb) hipLaunchKernelGGL(cb_rhs_kernel, dim3(nb), dim3(NT), 0, s, table);
}
void launch_cb_assemble(const CbStep* table, uint32_t nb, const CbMembers& m, uint32_t npp, uint32_t off, uint32_t ext, hipStream_t s) {
    if (nb) hipLaunchKernelGGL(cb_assemble_kernel, dim3(ext / 128, ext / 128, 4 * nb), dim3(256), 0, s, table, m, npp, off);
}
void launch_cb_post(const CbStep* table, uint32_t nb, const CbMembers& m, uint32_t nip, uint32_t npp, uint32_t outnp_max, hipStream_t s) {
    const uint32_t rows = outnp_max > nip ? outnp_max : nip;
    if (nb) hipLaunchKernelGGL(cb_post_kernel, dim3((rows + 255) / 256, outnp_max + 1, nb), dim3(256), 0, s, table, m, nip, npp, outnp_max);
}
void launch_cb_rhs_steps(const CbStep* table, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(cb_rhs_steps_kernel, dim3(n), dim3(NT), 0, s, table);
}

}  // namespace dnagpu
