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
#include "la_kernels.h"

namespace dnagpu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ----------------------------------------------------------------------------
// Tile GEMM:  C(it,jt) = alpha * sum_k opA(i,k) opB(k,j) + beta * C(it,jt)
//   A_KC = false: A(i,k) at A[i + k*lda]   (row-contiguous, "N")
//   A_KC = true : A(i,k) at A[k + i*lda]   (k-contiguous,  "T")
//   B_KC = false: B(k,j) at B[j + k*ldb]   ("T": B given as its transpose)
//   B_KC = true : B(k,j) at B[k + j*ldb]   ("N")
// 256 threads = 4 waves (2x2), each wave owns a 64x64 sub-tile = 4x4 MFMA
// 16x16 accumulators (128 VGPRs).  LDS is double buffered, one barrier per
// BK=16 slab.  LDS layouts are chosen so that the MFMA fragment reads
// (ds_read_b64, two 32-lane groups) are bank-conflict free:
//   R layout  [k][row]      row stride 144 doubles (k+1 lands 32 banks away)
//   P layout  [k/2][col][2] pair stride 258 doubles (32 lanes read 256 B contiguous)
// ----------------------------------------------------------------------------
// Two block-tile sizes share the code: TILE = 128 (4 waves x 64x64, the throughput shape, ~70 TFLOP/s)
// and TILE = 64 (4 waves x 32x32) for the small nodes of the recursion, where a launch has only a
// handful of 128-tiles and latency, not throughput, is what matters (4x the workgroups, 1/4 the k-loop time).
template <int TILE>
struct Geo {
    static constexpr int LDR = TILE + 16;       // R layout row stride: k+1 lands 32 banks away
    static constexpr int LDP = 2 * TILE + 2;    // P layout pair stride
    static constexpr int OPBUF = 16 * LDR;      // doubles per operand buffer (>= 8 * LDP)
    static constexpr int NQ = TILE / 32;        // 16-byte loads per thread per operand slab
    static constexpr int WT = TILE / 2;         // wave tile
    static constexpr int MI = WT / 16;          // MFMA tiles per wave per dimension
};

// Global -> register staging of one BK=16 operand slab.  The per-thread part of the address is loop invariant
// (a 32-bit byte offset, computed once); the slab position is wave-uniform and travels in the scalar base, so the
// main loop issues `global_load_dwordx4 v, v_off, s[base]` with no vector address arithmetic.
template <bool KC, int TILE>
__device__ __forceinline__ void stage_offsets(int ld, int tid, uint32_t (&off)[Geo<TILE>::NQ]) {
#pragma unroll
    for (int q = 0; q < Geo<TILE>::NQ; ++q) {
        int idx = tid + 256 * q;
        if (!KC) {
            int k = idx / (TILE / 2), r2 = idx % (TILE / 2);
            off[q] = (uint32_t)(k * ld + 2 * r2) * 8u;
        } else {
            int k2 = idx & 7, c = idx >> 3;
            off[q] = (uint32_t)(c * ld + 2 * k2) * 8u;
        }
    }
}

template <bool KC>
__device__ __forceinline__ const char* stage_base(const double* __restrict__ P, int ld, int r0, int k0) {
    return reinterpret_cast<const char*>(KC ? P + (size_t)r0 * ld + k0 : P + (size_t)k0 * ld + r0);
}

template <int TILE>
__device__ __forceinline__ void stage_load(const char* base, const uint32_t (&off)[Geo<TILE>::NQ], d2 (&g)[Geo<TILE>::NQ]) {
#pragma unroll
    for (int q = 0; q < Geo<TILE>::NQ; ++q) g[q] = *reinterpret_cast<const d2*>(base + off[q]);
}

template <bool KC, int TILE>
__device__ __forceinline__ void stage_store(double* buf, int tid, const d2 (&g)[Geo<TILE>::NQ]) {
#pragma unroll
    for (int q = 0; q < Geo<TILE>::NQ; ++q) {
        int idx = tid + 256 * q;
        if (!KC) {
            int k = idx / (TILE / 2), r2 = idx % (TILE / 2);
            *reinterpret_cast<d2*>(buf + k * Geo<TILE>::LDR + 2 * r2) = g[q];
        } else {
            int k2 = idx & 7, c = idx >> 3;
            *reinterpret_cast<d2*>(buf + k2 * Geo<TILE>::LDP + 2 * c) = g[q];
        }
    }
}

template <bool KC, int TILE>
__device__ __forceinline__ double frag_read(const double* buf, int kk, int rbase, int lane) {
    int k = kk * 4 + (lane >> 4);
    int r = rbase + (lane & 15);
    if (!KC) return buf[k * Geo<TILE>::LDR + r];
    return buf[(k >> 1) * Geo<TILE>::LDP + r * 2 + (k & 1)];
}

template <bool A_KC, bool B_KC, int TILE>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs a) {
    using G = Geo<TILE>;
    __shared__ __attribute__((aligned(16))) double lds[4 * G::OPBUF];
    // tile order table built on the host (tile_order.hip): workgroup b runs on XCD b%8,
    // every XCD walks its own work-balanced list of 2-D super-tiles (L2 locality).
    const uint32_t packed = a.order[blockIdx.x];
    if (packed == 0xffffffffu) return;
    const int it = (int)(packed >> 16), jt = (int)(packed & 0xffffu);

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

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int i0 = it * TILE, j0 = jt * TILE;

    d4 acc[G::MI][G::MI];
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < G::MI; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

    d2 ga[G::NQ], gb[G::NQ];
    uint32_t offa[G::NQ], offb[G::NQ];
    stage_offsets<A_KC, TILE>(a.lda, tid, offa);
    stage_offsets<B_KC, TILE>(a.ldb, tid, offb);
    const int nk = (kend - kbeg) / 16;

    // MFMA fragments, two register sets: while the 16 MFMAs of k-step kk run, the fragments of kk+1 are on their way
    double af[2][G::MI], bf[2][G::MI];
    auto read_frags = [&](const double* As, const double* Bs, int kk, int set) {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) af[set][mi] = frag_read<A_KC, TILE>(As, kk, wm * G::WT + mi * 16, lane);
#pragma unroll
        for (int ni = 0; ni < G::MI; ++ni) bf[set][ni] = frag_read<B_KC, TILE>(Bs, kk, wn * G::WT + ni * 16, lane);
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::MI; ++ni)
                // first operand indexes the result row (= j), second the result
                // column (= i = lane&15): stores become 128 B contiguous in i.
                acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[set][ni], af[set][mi], acc[mi][ni], 0, 0, 0);
    };

    if (nk > 0) {
        stage_load<TILE>(stage_base<A_KC>(a.A, a.lda, i0, kbeg), offa, ga);
        stage_load<TILE>(stage_base<B_KC>(a.B, a.ldb, j0, kbeg), offb, gb);
        stage_store<A_KC, TILE>(lds, tid, ga);
        stage_store<B_KC, TILE>(lds + G::OPBUF, tid, gb);
    }
    __syncthreads();
    if (nk > 0) read_frags(lds, lds + G::OPBUF, 0, 0);

    // One barrier per slab, placed BEFORE the last k-step: the slab boundary (barrier skew + LDS latency of the next
    // slab's first fragments) is covered by that k-step's 16 MFMAs instead of leaving the MFMA pipe idle.
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const double* As = lds + cur * 2 * G::OPBUF;
        const double* Bs = As + G::OPBUF;
        double* An = lds + (cur ^ 1) * 2 * G::OPBUF;
        const bool more = (t + 1 < nk);
        if (more) {
            stage_load<TILE>(stage_base<A_KC>(a.A, a.lda, i0, kbeg + (t + 1) * 16), offa, ga);
            stage_load<TILE>(stage_base<B_KC>(a.B, a.ldb, j0, kbeg + (t + 1) * 16), offb, gb);
        }
        read_frags(As, Bs, 1, 1);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(As, Bs, 2, 0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(As, Bs, 3, 1);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            stage_store<A_KC, TILE>(An, tid, ga);
            stage_store<B_KC, TILE>(An + G::OPBUF, tid, gb);
        }
        __syncthreads();
        if (more) read_frags(An, An + G::OPBUF, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // epilogue: acc[mi][ni][r] = C(i = i0+wm*WT+mi*16+(lane&15), j = j0+wn*WT+ni*16+(lane>>4)+4r)
    const bool mirror = a.mirror && (it != jt);
    double* cbase = a.C + (size_t)(j0 + wn * G::WT + (lane >> 4)) * a.ldc + i0 + wm * G::WT + (lane & 15);
    if (a.beta != 0.0) {
        // C tile read in batches of 8 independent loads before it is combined (not one load-wait per element)
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
#pragma unroll
            for (int n2 = 0; n2 < G::MI; n2 += 2) {
                double cold[2][4];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cold[ni][r] = cbase[(size_t)((n2 + ni) * 16 + 4 * r) * a.ldc + mi * 16];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mi][n2 + ni][r] = a.alpha * acc[mi][n2 + ni][r] + a.beta * cold[ni][r];
            }
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::MI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = a.alpha * acc[mi][ni][r];
    }
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < G::MI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[mi][ni][r];
                cbase[(size_t)(ni * 16 + 4 * r) * a.ldc + mi * 16] = v;
                if (mirror) {
                    int i = i0 + wm * G::WT + mi * 16 + (lane & 15);
                    int j = j0 + wn * G::WT + ni * 16 + (lane >> 4) + 4 * r;
                    a.C[(size_t)i * a.ldc + j] = v;
                }
            }
        }
    }
}

template <int TILE>
static void launch_gemm_t(const GemmArgs& a, int a_kc, int b_kc, hipStream_t s) {
    dim3 grid(a.grid), block(256);
    if (!a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<false, false, TILE>), grid, block, 0, s, a);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<false, true, TILE>), grid, block, 0, s, a);
    else if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<true, true, TILE>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gemm_f64_kernel<true, false, TILE>), grid, block, 0, s, a);
}

void launch_gemm(const GemmArgs& a, int a_kc, int b_kc, hipStream_t s) {
    if (a.grid <= 0) return;
    if (a.tile == 64)
        launch_gemm_t<64>(a, a_kc, b_kc, s);
    else
        launch_gemm_t<128>(a, a_kc, b_kc, s);
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
