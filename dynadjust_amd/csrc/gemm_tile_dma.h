// The 128 x 128 fp64 tile product of the throughput kernel as a device function: one workgroup of 64 * WAVES threads computes
//   C(i0.., j0..) = alpha * sum_{k in [kbeg, kend)} opA(i, k) opB(k, j) + beta * C
// with v_mfma_f64_16x16x4_f64, operands global -> LDS by LDS-DMA (global_load_lds_dwordx4).  Used by gemm_f64_dma_kernel (one launch
// per product, la_kernels.hip).  Replaces the dgemm / dsyrk / dtrsm calls inside dpotrf / dpotri of
// matrix_2d::cholesky_inverse (dynadjust/include/math/dnamatrix_contiguous.cpp:952-1020).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dnagpu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Shapes: TILE = 128 / 8 waves (throughput), TILE = 128 / 4 waves (kept for comparison, DNAGPU_GEMM_WAVES=4), and
// TILE = 64 / 4 waves x 32x32 for the small nodes of the recursion, where a launch has only a handful of
// 128-tiles and latency, not throughput, is what matters (4x the workgroups, 1/4 the k-loop time).
template <int TILE, int WAVES>
struct Geo {
    static constexpr int NT = 64 * WAVES;       // threads per workgroup
    static constexpr int LDR = TILE + 16;       // R layout row stride: k+1 lands 32 banks away
    static constexpr int LDP = 2 * TILE + 2;    // P layout pair stride
    static constexpr int OPBUF = 16 * LDR;      // doubles per operand buffer (>= 8 * LDP)
    static constexpr int NQ = TILE * 8 / NT;    // 16-byte loads per thread per operand slab
    static constexpr int WTM = TILE / 2;        // wave tile rows
    static constexpr int WTN = TILE / (WAVES / 2);   // wave tile columns
    static constexpr int MI = WTM / 16;         // MFMA tiles per wave along i
    static constexpr int NI = WTN / 16;         // MFMA tiles per wave along j
};

template <bool KC>
__device__ __forceinline__ const char* stage_base(const double* __restrict__ P, int ld, int r0, int k0) {
    return reinterpret_cast<const char*>(KC ? P + (size_t)r0 * ld + k0 : P + (size_t)k0 * ld + r0);
}

template <bool KC, int TILE, int WAVES>
__device__ __forceinline__ double frag_read(const double* buf, int kk, int rbase, int lane) {
    int k = kk * 4 + (lane >> 4);
    int r = rbase + (lane & 15);
    if (!KC) return buf[k * Geo<TILE, WAVES>::LDR + r];
    return buf[(k >> 1) * Geo<TILE, WAVES>::LDP + r * 2 + (k & 1)];
}

// ---- global -> LDS without a register round trip (global_load_lds_dwordx4, TILE = 128 only) ---------------------
// One wave instruction moves 64 lanes x 16 B and lands them CONTIGUOUSLY (lane order) at a wave-uniform LDS address,
// so the layouts are chosen such that every wave instruction fills one contiguous 1 KiB piece:
//   R layout (row-contiguous operand): piece = one k-row of 128 doubles, rows LDR apart (as before);
//   S layout (k-contiguous operand)  : piece = "chunk" of 8 columns x 16 k.  Inside a chunk the 16-byte unit of column c
//       (0..7) and k-pair k2 (0..7) sits at position c*8 + ((k2 + (c>>1) + 4*(chunk&1)) & 7): lanes 8c..8c+7 still read
//       one 128 B line of column c (coalesced), and the rotation makes the MFMA fragment reads (16 columns x 4 k per
//       ds_read_b64) hit 32 different bank pairs per half wave.
template <bool KC, int TILE, int WAVES>
__device__ __forceinline__ void dma_offsets(int ld, int wave, int lane, uint32_t (&off)[Geo<TILE, WAVES>::NQ]) {
#pragma unroll
    for (int q = 0; q < Geo<TILE, WAVES>::NQ; ++q) {
        const int piece = wave + WAVES * q;          // k-row (R) or chunk (S), 0..15
        if (!KC) {
            off[q] = (uint32_t)(piece * ld + 2 * lane) * 8u;
        } else {
            const int c = lane >> 3, x = lane & 7, k2 = (x - (c >> 1) - 4 * (piece & 1)) & 7;
            off[q] = (uint32_t)((piece * 8 + c) * ld + 2 * k2) * 8u;
        }
    }
}

template <bool KC, int TILE, int WAVES>
__device__ __forceinline__ void dma_issue(const char* base, const uint32_t (&off)[Geo<TILE, WAVES>::NQ], double* buf, int wave) {
#pragma unroll
    for (int q = 0; q < Geo<TILE, WAVES>::NQ; ++q) {
        const int piece = wave + WAVES * q;
        double* dst = buf + (KC ? piece * 128 : piece * Geo<TILE, WAVES>::LDR);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off[q]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// fragment element (row/col r, k) of the S layout
template <int TILE, int WAVES>
__device__ __forceinline__ double frag_read_s(const double* buf, int kk, int rbase, int lane) {
    const int k = kk * 4 + (lane >> 4);
    const int r = rbase + (lane & 15);
    const int chunk = r >> 3, c = r & 7;
    const int x = ((k >> 1) + (c >> 1) + 4 * (chunk & 1)) & 7;
    return buf[chunk * 128 + (c * 8 + x) * 2 + (k & 1)];
}

// One 128 x 128 tile of C at (i0, j0), k over [kbeg, kend) (multiples of 16, an even number of 16-slabs: the callers' ranges are
// multiples of 128), walked downwards when `down` (ranges that END at a k all tiles of a launch share: the workgroups that share
// operand panels in their XCD's L2 then start together and stay in step).  lds0 / lds1: the two operand buffers, 2 * OPBUF doubles
// each, DISTINCT __shared__ objects of the calling kernel (the compiler then knows that a DMA into one never aliases the fragment
// reads of the other and does not serialise them behind vmcnt).  The caller separates consecutive tiles of one workgroup by a barrier.
template <bool A_KC, bool B_KC, int WAVES>
__device__ __forceinline__ void dma_tile_product(const double* A, int lda, const double* B, int ldb,
                                                 double* C, int ldc, int i0, int j0, int kbeg, int kend, bool down,
                                                 double alpha, double beta, bool mirror, double* lds0, double* lds1) {
    constexpr int TILE = 128;
    using G = Geo<TILE, WAVES>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    uint32_t offa[G::NQ], offb[G::NQ];
    dma_offsets<A_KC, TILE, WAVES>(lda, wave, lane, offa);
    dma_offsets<B_KC, TILE, WAVES>(ldb, wave, lane, offb);

        d4 acc[G::MI][G::NI];
    #pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
    #pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

        const int nk = (kend - kbeg) / 16;

        double af[2][G::MI], bf[2][G::NI];
        auto read_frags = [&](const double* As, const double* Bs, int kk, int set) {
    #pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
                af[set][mi] = A_KC ? frag_read_s<TILE, WAVES>(As, kk, wm * G::WTM + mi * 16, lane)
                                   : frag_read<false, TILE, WAVES>(As, kk, wm * G::WTM + mi * 16, lane);
    #pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
                bf[set][ni] = B_KC ? frag_read_s<TILE, WAVES>(Bs, kk, wn * G::WTN + ni * 16, lane)
                                   : frag_read<false, TILE, WAVES>(Bs, kk, wn * G::WTN + ni * 16, lane);
        };
        auto mfmas = [&](int set) {
    #pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
    #pragma unroll
                for (int ni = 0; ni < G::NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[set][ni], af[set][mi], acc[mi][ni], 0, 0, 0);
        };
        // Triangular ranges that END at a common k (k >= i, k >= j) are walked downwards: the workgroups of a super-tile,
        // which share operand panels in their XCD's L2, then start together at the common end and stay in step, instead of
        // each starting at its own first k and drifting apart by (tile distance) x 128 for the whole run.
            auto issue = [&](int t, double* buf) {
            const int k0 = down ? kend - (t + 1) * 16 : kbeg + t * 16;
            dma_issue<A_KC, TILE, WAVES>(stage_base<A_KC>(A, lda, i0, k0), offa, buf, wave);
            dma_issue<B_KC, TILE, WAVES>(stage_base<B_KC>(B, ldb, j0, k0), offb, buf + G::OPBUF, wave);
        };

        if (nk > 0) issue(0, lds0);
        __syncthreads();
        if (nk > 0) read_frags(lds0, lds0 + G::OPBUF, 0, 0);

        constexpr int NM = G::MI * G::NI;                 // MFMAs per k-step
        constexpr int NL = 2 * G::NQ;                     // DMA instructions per slab
        constexpr int NR = G::MI + G::NI;                 // fragment reads per k-step (before ds_read2 merging)
        // slab t lives in `cur`; the DMA for slab t+1 goes to `nxt` (free since the barrier of slab t-1); one barrier per
        // slab, placed before the last k-step (see gemm_f64_kernel)
        auto slab = [&](int t, const double* cur, double* nxt, auto more_tag) {
            constexpr bool more = decltype(more_tag)::value;
            const double* As = cur;
            const double* Bs = cur + G::OPBUF;
            if (more) issue(t + 1, nxt);
            read_frags(As, Bs, 1, 1);
            mfmas(0);
    #pragma unroll
            for (int g = 0; g < NL; ++g) {
                if (more) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // 1 VMEM (LDS-DMA)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, NM / NL, 0);      // MFMAs
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(As, Bs, 2, 0);
            mfmas(1);
    #pragma unroll
            for (int g = 0; g < NR / 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NM / (NR / 2), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            read_frags(As, Bs, 3, 1);
            mfmas(0);
    #pragma unroll
            for (int g = 0; g < NR / 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NM / (NR / 2), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                                  // (waits for this wave's DMA: vmcnt(0))
            if (more) read_frags(nxt, nxt + G::OPBUF, 0, 0);
            mfmas(1);
    #pragma unroll
            for (int g = 0; g < NR / 2; ++g) {
                if (more) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NM / (NR / 2), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // k ranges are multiples of 128, so the slab count is even: slabs go in (lds0, lds1) pairs, the last pair peeled
        for (int t = 0; t + 2 < nk; t += 2) {
            slab(t, lds0, lds1, std::true_type{});
            slab(t + 1, lds1, lds0, std::true_type{});
        }
        if (nk > 0) {
            slab(nk - 2, lds0, lds1, std::true_type{});
            slab(nk - 1, lds1, lds0, std::false_type{});
        }

        // epilogue (as gemm_f64_kernel)
            double* cbase = C + (size_t)(j0 + wn * G::WTN + (lane >> 4)) * ldc + i0 + wm * G::WTM + (lane & 15);
        if (beta != 0.0) {
    #pragma unroll
            for (int mi = 0; mi < G::MI; ++mi) {
    #pragma unroll
                for (int n2 = 0; n2 < G::NI; n2 += 2) {
                    double cold[2][4];
    #pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) cold[ni][r] = cbase[(size_t)((n2 + ni) * 16 + 4 * r) * ldc + mi * 16];
    #pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
    #pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mi][n2 + ni][r] = alpha * acc[mi][n2 + ni][r] + beta * cold[ni][r];
                }
            }
        } else {
    #pragma unroll
            for (int mi = 0; mi < G::MI; ++mi)
    #pragma unroll
                for (int ni = 0; ni < G::NI; ++ni)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mi][ni][r] = alpha * acc[mi][ni][r];
        }
    #pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
    #pragma unroll
            for (int ni = 0; ni < G::NI; ++ni) {
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = acc[mi][ni][r];
                    cbase[(size_t)(ni * 16 + 4 * r) * ldc + mi * 16] = v;
                    if (mirror) {
                        int i = i0 + wm * G::WTM + mi * 16 + (lane & 15);
                        int j = j0 + wn * G::WTN + ni * 16 + (lane >> 4) + 4 * r;
                        C[(size_t)i * ldc + j] = v;
                    }
                }
            }
        }
}

}  // namespace dnagpu
