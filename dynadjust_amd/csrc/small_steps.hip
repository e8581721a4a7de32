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

// y[i] = base[i] + sign * sum_{j < cols, (LOWER: j <= i)} A[i + j * lda] * x[j],  i < rows.   x, base, y: LDS (y may be base, not x);
// part: NW * 64 doubles of LDS.  Rows in groups of 64 (a lane per row: coalesced columns), the waves left over split the columns
// (interleaved), eight independent accumulators per lane, the column slices added in a fixed order: deterministic.
template <bool LOWER>
__device__ void wg_matvec(const double* __restrict__ A, uint32_t lda, uint32_t rows, uint32_t cols, const double* x, const double* base, double sign,
                          double* y, double* part) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
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

}  // namespace

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

}  // namespace dnagpu
