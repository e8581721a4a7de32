// Dense fp64 linear-algebra kernels for gfx950 (MI355X): the arithmetic behind
// dna_adjust::Solve -> FormInverseVarianceMatrix -> matrix_2d::cholesky_inverse
// (reference: dynadjust/dynadjust/dnaadjust/dnaadjust.cpp:6586-6647, 8472-8517;
//  dynadjust/include/math/dnamatrix_contiguous.cpp:952-1020 = dpotrf + dpotri).
//
// Layout in HBM: every symmetric matrix lives in a full square column-major
// buffer of order np = ceil(n/128)*128 (ld = np).  The padding carries an
// identity so that every kernel works on whole 128x128 tiles and never needs
// edge predication.  The reference's packed-lower exchange format
// (matrix_2d::packed_index, dnamatrix_contiguous.hpp:363) only exists at the
// host boundary (pack/unpack kernels below).
//
// The inverse is a recursive blocked algorithm whose flops all go through ONE
// tile kernel (gemm_f64) built on v_mfma_f64_16x16x4_f64:
//   node(o,s):  node(left half)                      -> X11 = L11^-1
//               W21  = A21 * X11^T        (NT, k <= j)   == L21
//               A22 -= W21 * W21^T        (NT, lower)
//               node(right half)                     -> X22
//               T21  = W21 * X11          (NN, k >= j)
//               X21  = -X22 * T21         (NN, k <= i)
//   lauum:      Ninv = X^T X              (TN, k >= i, mirrored to both triangles)
// Flop count n^3/3 + n^3/3 + n^3/3 = the reference's dpotrf + dpotri.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "la_kernels.h"
#include "gemm_tile_dma.h"
#include "gemm_tile_reg.h"

namespace dnagpu {

// ----------------------------------------------------------------------------
// Tile GEMM:  C(it,jt) = alpha * sum_k opA(i,k) opB(k,j) + beta * C(it,jt)
//   A_KC = false: A(i,k) at A[i + k*lda]   (row-contiguous, "N")
//   A_KC = true : A(i,k) at A[k + i*lda]   (k-contiguous,  "T")
//   B_KC = false: B(k,j) at B[j + k*ldb]   ("T": B given as its transpose)
//   B_KC = true : B(k,j) at B[k + j*ldb]   ("N")
// A workgroup owns a TILE x TILE block of C; its WAVES waves form a 2 x (WAVES/2) grid, each wave a
// (TILE/2) x (TILE/(WAVES/2)) sub-tile of 16x16 MFMA accumulators.  LDS is double buffered, one barrier per
// BK=16 slab.  LDS layouts are chosen so that the MFMA fragment reads (ds_read_b64, two 32-lane groups) are
// bank-conflict free:
//   R layout  [k][row]      row stride TILE+16 doubles (k+1 lands 32 banks away)
//   P layout  [k/2][col][2] pair stride 2*TILE+2 doubles (32 lanes read 256 B contiguous)
//
// Occupancy is the design parameter (probe: tools/probes/mfma_f64_peak.hip).  One wave issues
// v_mfma_f64_16x16x4_f64 back to back at only 46 % of the pipe's rate (36 of 78.6 TFLOP/s with one wave per SIMD);
// two waves per SIMD reach 99 % -- but only while BOTH are issuing.  With 4 waves x 64x64 (128 accumulator
// VGPRs, 2 waves per SIMD) every barrier / LDS / address instruction of one wave halves the SIMD's MFMA rate and
// the kernel saturates at 89 % busy.  The throughput shape is therefore 8 waves x 64x32 (64 accumulator VGPRs,
// <= 128 VGPRs per wave, 2 workgroups per CU = 4 waves per SIMD): any three of the four can keep the pipe full.
// ----------------------------------------------------------------------------

// One TILE x TILE tile of C (it, jt in units of TILE) of a launch: the k range from the launch's kmode, then reg_tile_product
// (gemm_tile_reg.h); `lds`: 4 * OPBUF doubles.
template <bool A_KC, bool B_KC, int TILE, int WAVES, class Args>
__device__ __forceinline__ void gemm_tile_body(const Args& a, const int it, const int jt, double* lds, long long dA = 0, long long dB = 0,
                                               long long dC = 0) {
    using G = Geo<TILE, WAVES>;
    // triangular operands restrict the k range in units of the 128-wide blocks of the recursion
    const int bi = (it * TILE) / 128, bj = (jt * TILE) / 128;
    int kbeg = 0, kend = a.K;
    switch (a.kmode) {
        case KM_LE_J: kend = (bj + 1) * 128; break;
        case KM_GE_J: kbeg = bj * 128; break;
        case KM_LE_I: kend = (bi + 1) * 128; break;
        case KM_GE_I: kbeg = bi * 128; break;
        default: break;
    }
    if (kend > a.K) kend = a.K;
    reg_tile_product<A_KC, B_KC, TILE, WAVES>(a.A + dA, a.lda, a.B + dB, a.ldb, a.C + dC, a.ldc, it * TILE, jt * TILE, kbeg, kend, a.alpha, a.beta,
                                              a.mirror && (it != jt), lds, lds + 2 * G::OPBUF);
}

template <bool A_KC, bool B_KC, int TILE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 2) void gemm_f64_kernel(GemmArgs a) {   // 2nd argument: waves per SIMD (2 workgroups per CU)
    using G = Geo<TILE, WAVES>;
    __shared__ __attribute__((aligned(16))) double lds[4 * G::OPBUF];
    // tile order table built on the host (tile_order.hip): workgroup b runs on XCD b%8,
    // every XCD walks its own work-balanced list of 2-D super-tiles (L2 locality).
    const uint32_t packed = a.order[blockIdx.x];
    if (packed == 0xffffffffu) return;
    long long dA = 0, dB = 0, dC = 0;
    if (a.nb > 1) {     // a batched launch: this workgroup's member (uniform: scalar loads from the kernel arguments)
        const int b = (int)blockIdx.y;
        dA = a.dA[b];
        dB = a.dB[b];
        dC = a.dC[b];
    }
    gemm_tile_body<A_KC, B_KC, TILE, WAVES>(a, (int)(packed >> 16), (int)(packed & 0xffffu), lds, dA, dB, dC);
}

// The throughput kernel: TILE = 128, operands staged with LDS-DMA (no staging registers, no ds_write), two separate
// LDS buffers (distinct __shared__ objects, so that the compiler knows a DMA into one never aliases the fragment
// reads of the other and does not serialise them behind vmcnt).  The tile itself: dma_tile_product (gemm_tile_dma.h).
template <bool A_KC, bool B_KC, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 2) void gemm_f64_dma_kernel(GemmArgs a) {
    constexpr int TILE = 128;
    using G = Geo<TILE, WAVES>;
    __shared__ __attribute__((aligned(16))) double lds0[2 * G::OPBUF];
    __shared__ __attribute__((aligned(16))) double lds1[2 * G::OPBUF];
    long long dA = 0, dB = 0, dC = 0;
    if (a.nb > 1) {     // a batched launch: this workgroup's member (uniform: scalar loads from the kernel arguments)
        const int b = (int)blockIdx.y;
        dA = a.dA[b];
        dB = a.dB[b];
        dC = a.dC[b];
    }
    const uint32_t packed = a.order[blockIdx.x];
    if (packed == 0xffffffffu) return;
    const int it = (int)(packed >> 16), jt = (int)(packed & 0x7fffu);
    int kbeg = 0, kend = a.K;
    switch (a.kmode) {
        case KM_LE_J: kend = (jt + 1) * 128; break;
        case KM_GE_J: kbeg = jt * 128; break;
        case KM_LE_I: kend = (it + 1) * 128; break;
        case KM_GE_I: kbeg = it * 128; break;
        default: break;
    }
    if (kend > a.K) kend = a.K;
    // Triangular ranges that END at a common k (k >= i, k >= j) are walked downwards: the workgroups of a super-tile,
    // which share operand panels in their XCD's L2, then start together at the common end and stay in step, instead of
    // each starting at its own first k and drifting apart by (tile distance) x 128 for the whole run.
    const bool down = a.kmode == KM_GE_J || a.kmode == KM_GE_I;
    dma_tile_product<A_KC, B_KC, WAVES>(a.A + dA, a.lda, a.B + dB, a.ldb, a.C + dC, a.ldc, it * TILE, jt * TILE, kbeg, kend, down, a.alpha, a.beta,
                                        a.mirror && (it != jt), lds0, lds1);
}


template <int WAVES>
static void launch_gemm_dma(const GemmArgs& a, int a_kc, int b_kc, hipStream_t s) {
    dim3 grid(a.grid, a.nb), block(64 * WAVES);
    if (!a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f64_dma_kernel<false, false, WAVES>), grid, block, 0, s, a);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_dma_kernel<false, true, WAVES>), grid, block, 0, s, a);
    else if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_dma_kernel<true, true, WAVES>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gemm_f64_dma_kernel<true, false, WAVES>), grid, block, 0, s, a);
}

template <int TILE, int WAVES>
static void launch_gemm_t(const GemmArgs& a, int a_kc, int b_kc, hipStream_t s) {
    dim3 grid(a.grid, a.nb), block(64 * WAVES);
    if (!a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<false, false, TILE, WAVES>), grid, block, 0, s, a);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<false, true, TILE, WAVES>), grid, block, 0, s, a);
    else if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<true, true, TILE, WAVES>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gemm_f64_kernel<true, false, TILE, WAVES>), grid, block, 0, s, a);
}

void launch_gemm(const GemmArgs& a, int a_kc, int b_kc, hipStream_t s) {
    if (a.grid <= 0) return;
    if (a.tile == 64)
        launch_gemm_t<64, 4>(a, a_kc, b_kc, s);
    else if (a.tile == 32)
        launch_gemm_t<32, 4>(a, a_kc, b_kc, s);
    else
        launch_gemm_dma<4>(a, a_kc, b_kc, s);
}

// ----------------------------------------------------------------------------
// pack / unpack between the reference's packed-lower column-major layout
// (index j*n - j(j-1)/2 + (i-j), dnamatrix_contiguous.hpp:363) and the padded
// full-square device layout.  unpack also writes the identity padding.
// ----------------------------------------------------------------------------
__global__ void unpack_lower_kernel(const double* __restrict__ ap, double* __restrict__ F, uint32_t n, uint32_t np) {
    uint32_t j = blockIdx.x;  // column
    size_t colbase = (size_t)j * n - (size_t)j * (j > 0 ? j - 1 : 0) / 2;
    for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) {
        double v;
        if (j < n && i < n)
            v = (i >= j) ? ap[colbase + (i - j)] : 0.0;
        else
            v = (i == j) ? 1.0 : 0.0;
        F[(size_t)j * np + i] = v;
    }
}

__global__ void pack_lower_kernel(const double* __restrict__ F, double* __restrict__ ap, uint32_t n, uint32_t np) {
    uint32_t j = blockIdx.x;
    size_t colbase = (size_t)j * n - (size_t)j * (j > 0 ? j - 1 : 0) / 2;
    for (uint32_t i = j + threadIdx.x; i < n; i += blockDim.x) ap[colbase + (i - j)] = F[(size_t)j * np + i];
}

void launch_unpack_lower(const double* ap, double* F, uint32_t n, uint32_t np, hipStream_t s) {
    hipLaunchKernelGGL(unpack_lower_kernel, dim3(np), dim3(256), 0, s, ap, F, n, np);
}
void launch_pack_lower(const double* F, double* ap, uint32_t n, uint32_t np, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(pack_lower_kernel, dim3(n), dim3(256), 0, s, F, ap, n, np);
}

// zero the whole buffer and put 1.0 on the padded part of the diagonal
__global__ void init_padded_kernel(double* __restrict__ F, uint32_t n, uint32_t np) {
    uint32_t j = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) F[(size_t)j * np + i] = (i == j && j >= n) ? 1.0 : 0.0;
}
// the same for the padding only (rows and columns n .. np - 1): for a matrix whose n x n part is about to be written in full
__global__ void init_padding_kernel(double* __restrict__ F, uint32_t n, uint32_t np) {
    uint32_t j = blockIdx.x;
    for (uint32_t i = (j >= n ? 0 : n) + threadIdx.x; i < np; i += blockDim.x) F[(size_t)j * np + i] = (i == j) ? 1.0 : 0.0;
}
void launch_init_padding(double* F, uint32_t n, uint32_t np, hipStream_t s) {
    if (np > n) hipLaunchKernelGGL(init_padding_kernel, dim3(np), dim3(128), 0, s, F, n, np);
}
void launch_init_padded(double* F, uint32_t n, uint32_t np, hipStream_t s) {
    hipLaunchKernelGGL(init_padded_kernel, dim3(np), dim3(256), 0, s, F, n, np);
}

// ----------------------------------------------------------------------------
// scale_normals_to_unity (dnaadjust.cpp:6614-6645, scale_symmetric_diagonal
// dnamatrix_contiguous.cpp:1145): s_i = 1/sqrt(N_ii);  N <- S N S (lower or full)
// ----------------------------------------------------------------------------
__global__ void diag_rsqrt_kernel(const double* __restrict__ F, double* __restrict__ s, uint32_t n, uint32_t np) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) s[i] = (i < n) ? 1.0 / sqrt(F[(size_t)i * np + i]) : 1.0;
}
__global__ void scale_sym_kernel(double* __restrict__ F, const double* __restrict__ s, uint32_t n, uint32_t np, int lower_only) {
    uint32_t j = blockIdx.x;
    double sj = s[j];
    uint32_t ibeg = lower_only ? j : 0;
    for (uint32_t i = ibeg + threadIdx.x; i < n; i += blockDim.x) F[(size_t)j * np + i] *= s[i] * sj;
}
void launch_diag_rsqrt(const double* F, double* s, uint32_t n, uint32_t np, hipStream_t st) {
    hipLaunchKernelGGL(diag_rsqrt_kernel, dim3((np + 255) / 256), dim3(256), 0, st, F, s, n, np);
}
void launch_scale_sym(double* F, const double* s, uint32_t n, uint32_t np, int lower_only, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(scale_sym_kernel, dim3(n), dim3(256), 0, st, F, s, n, np, lower_only);
}

// copy the lower triangle into the upper one (tile transposes through LDS)
__global__ __launch_bounds__(256) void symmetrize_kernel(double* __restrict__ F, uint32_t np) {
    __shared__ double t[32][33];
    uint32_t bi = blockIdx.x, bj = blockIdx.y;
    if (bi <= bj) return;  // strictly-lower 32x32 tiles only; diagonal tiles handled below
    uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (uint32_t r = ty; r < 32; r += 8) t[r][tx] = F[(size_t)(bj * 32 + r) * np + bi * 32 + tx];  // t[col][row]
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8) F[(size_t)(bi * 32 + r) * np + bj * 32 + tx] = t[tx][r];
}
__global__ __launch_bounds__(256) void symmetrize_diag_kernel(double* __restrict__ F, uint32_t np) {
    uint32_t b = blockIdx.x;
    uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (uint32_t r = ty; r < 32; r += 8) {
        uint32_t i = b * 32 + tx, j = b * 32 + r;  // element (i, j)
        if (i > j) F[(size_t)i * np + j] = F[(size_t)j * np + i];
    }
}
void launch_symmetrize(double* F, uint32_t n, uint32_t np, hipStream_t s) {
    (void)n;
    uint32_t nb = np / 32;
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, s, F, np);
    hipLaunchKernelGGL(symmetrize_diag_kernel, dim3(nb), dim3(256), 0, s, F, np);
}

// ----------------------------------------------------------------------------
// y = F x for a full symmetric F (both triangles valid), deterministic:
// phase 1: partial[c][i] = sum over column chunk c;  phase 2: y_i = sum_c partial.
// HBM-bound: reads np*n*8 bytes once.  (reference: multiply_sym -> dspmv/dsymm,
// dnamatrix_contiguous.cpp:1471-1510.)
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void symv_partial_kernel(const double* __restrict__ F, const double* __restrict__ x,
                                                           double* __restrict__ part, uint32_t n, uint32_t np, uint32_t cols_per_chunk) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t c = blockIdx.y;
    uint32_t j0 = c * cols_per_chunk;
    uint32_t j1 = j0 + cols_per_chunk;
    if (j1 > n) j1 = n;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    if (i < np) {
        uint32_t j = j0;
        for (; j + 4 <= j1; j += 4) {
            acc0 += F[(size_t)j * np + i] * x[j];
            acc1 += F[(size_t)(j + 1) * np + i] * x[j + 1];
            acc2 += F[(size_t)(j + 2) * np + i] * x[j + 2];
            acc3 += F[(size_t)(j + 3) * np + i] * x[j + 3];
        }
        for (; j < j1; ++j) acc0 += F[(size_t)j * np + i] * x[j];
        part[(size_t)c * np + i] = (acc0 + acc1) + (acc2 + acc3);
    }
}
__global__ void symv_reduce_kernel(const double* __restrict__ part, double* __restrict__ y, uint32_t n, uint32_t np, uint32_t nchunks) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (uint32_t c = 0; c < nchunks; ++c) s += part[(size_t)c * np + i];
    y[i] = s;
}
void launch_symv(const double* F, const double* x, double* y, double* part, uint32_t n, uint32_t np, uint32_t nchunks, hipStream_t st) {
    if (n == 0) return;
    uint32_t cpc = (n + nchunks - 1) / nchunks;
    hipLaunchKernelGGL(symv_partial_kernel, dim3(np / 256 + (np % 256 ? 1 : 0), nchunks), dim3(256), 0, st, F, x, part, n, np, cpc);
    hipLaunchKernelGGL(symv_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, part, y, n, np, nchunks);
}

}  // namespace dnagpu
