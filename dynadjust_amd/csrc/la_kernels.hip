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

namespace dnagpu {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

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

// Global -> register staging of one BK=16 operand slab.  The per-thread part of the address is loop invariant
// (a 32-bit byte offset, computed once); the slab position is wave-uniform and travels in the scalar base, so the
// main loop has no vector address arithmetic beyond one 64-bit add per load.
template <bool KC, int TILE, int WAVES>
__device__ __forceinline__ void stage_offsets(int ld, int tid, uint32_t (&off)[Geo<TILE, WAVES>::NQ]) {
#pragma unroll
    for (int q = 0; q < Geo<TILE, WAVES>::NQ; ++q) {
        int idx = tid + Geo<TILE, WAVES>::NT * q;
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

template <int NQ>
__device__ __forceinline__ void stage_load(const char* base, const uint32_t (&off)[NQ], d2 (&g)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) g[q] = *reinterpret_cast<const d2*>(base + off[q]);
}

template <bool KC, int TILE, int WAVES>
__device__ __forceinline__ void stage_store(double* buf, int tid, const d2 (&g)[Geo<TILE, WAVES>::NQ]) {
    using G = Geo<TILE, WAVES>;
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
        int idx = tid + G::NT * q;
        if (!KC) {
            int k = idx / (TILE / 2), r2 = idx % (TILE / 2);
            *reinterpret_cast<d2*>(buf + k * G::LDR + 2 * r2) = g[q];
        } else {
            int k2 = idx & 7, c = idx >> 3;
            *reinterpret_cast<d2*>(buf + k2 * G::LDP + 2 * c) = g[q];
        }
    }
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

// One TILE x TILE tile of C (it, jt in units of TILE) by the calling workgroup of 64 * WAVES threads; `lds`: 4 * OPBUF doubles.
// A: the launch's arguments (order / grid unused here).  The caller separates consecutive tiles of one workgroup by a barrier.
template <bool A_KC, bool B_KC, int TILE, int WAVES, class Args>
__device__ __forceinline__ void gemm_tile_body(const Args& a, const int it, const int jt, double* lds) {
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

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int i0 = it * TILE, j0 = jt * TILE;

    d4 acc[G::MI][G::NI];
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};

    // Staging registers: a ring of RS slabs.  A 64-tile slab is 16 MFMAs per wave (0.43 us at the MFMA rate) against a
    // global-load latency of 0.7-1 us: three slabs in flight (12 more registers per operand) instead of one keep the loop off
    // the memory latency.  Measured gain: small (36 workgroups, K = 512: 25 -> 23 us; 136 workgroups, K = 1024: 49 -> 44 us) --
    // these launches are bound by the MFMA rate of the few CUs they occupy (13.7 us of the 23), not by the loads.
    // Slab s lives in ring slot s % RS from its load (issued at the start of slab s - RS) to its LDS store (during slab s - 1).
    constexpr int RS = (TILE == 64) ? 3 : 1;
    d2 ga[RS][G::NQ], gb[RS][G::NQ];
    uint32_t offa[G::NQ], offb[G::NQ];
    stage_offsets<A_KC, TILE, WAVES>(a.lda, tid, offa);
    stage_offsets<B_KC, TILE, WAVES>(a.ldb, tid, offb);
    const int nk = (kend - kbeg) / 16;
    // (slabs beyond the last one are clamped to it: loaded again, stored to the idle buffer, never used -- the slab body stays
    // branch free)
    auto load_slab = [&](int sl, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int k0 = kbeg + (sl < nk ? sl : nk - 1) * 16;
        stage_load<G::NQ>(stage_base<A_KC>(a.A, a.lda, i0, k0), offa, ga[slot]);
        stage_load<G::NQ>(stage_base<B_KC>(a.B, a.ldb, j0, k0), offb, gb[slot]);
    };

    // MFMA fragments, two register sets: while the MFMAs of k-step kk run, the fragments of kk+1 are on their way
    double af[2][G::MI], bf[2][G::NI];
    auto read_frags = [&](const double* As, const double* Bs, int kk, int set) {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) af[set][mi] = frag_read<A_KC, TILE, WAVES>(As, kk, wm * G::WTM + mi * 16, lane);
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) bf[set][ni] = frag_read<B_KC, TILE, WAVES>(Bs, kk, wn * G::WTN + ni * 16, lane);
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
                // first operand indexes the result row (= j), second the result
                // column (= i = lane&15): stores become 128 B contiguous in i.
                acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[set][ni], af[set][mi], acc[mi][ni], 0, 0, 0);
    };

    if (nk > 0) {
        load_slab(0, std::integral_constant<int, 0>{});
        if (RS > 1) load_slab(1, std::integral_constant<int, 1 % RS>{});
        if (RS > 2) load_slab(2, std::integral_constant<int, 2 % RS>{});
        stage_store<A_KC, TILE, WAVES>(lds, tid, ga[0]);
        stage_store<B_KC, TILE, WAVES>(lds + G::OPBUF, tid, gb[0]);
    }
    __syncthreads();
    if (nk > 0) read_frags(lds, lds + G::OPBUF, 0, 0);

    // One barrier per slab, placed BEFORE the last k-step: the slab boundary (barrier skew + LDS latency of the next
    // slab's first fragments) is covered by that k-step's MFMAs.
    //
    // Issue order inside each k-step (sched_group_barrier): the global loads, fragment reads and LDS stores are spread
    // between the MFMAs instead of being issued in clusters.  A cluster of 8 global_load_dwordx4 at the top of the slab
    // alone costs 7 % of the MFMA rate (tools/probes/mfma_f64_feed.hip: 71.6 -> 66.1 TFLOP/s; spread out: 69.1).
    // The slab body is branch free (the last slab, which stages nothing, is peeled) so that the scheduler can do that.
    constexpr int NM = G::MI * G::NI;                 // MFMAs per k-step
    constexpr int NL = 2 * G::NQ;                     // global loads / LDS stores per slab
    constexpr int NR = G::MI + G::NI;                 // fragment reads per k-step (before ds_read2 merging)
    auto slab = [&](int t, auto slot_tag) {
        constexpr bool more = true;
        constexpr int slot = decltype(slot_tag)::value;      // t % RS: free since slab t went to LDS; slab t + RS moves in
        constexpr int nxt = (slot + 1) % RS;                 // slab t + 1: goes to LDS during this slab
        const int cur = t & 1;
        const double* As = lds + cur * 2 * G::OPBUF;
        const double* Bs = As + G::OPBUF;
        double* An = lds + (cur ^ 1) * 2 * G::OPBUF;
        load_slab(t + RS, slot_tag);
        read_frags(As, Bs, 1, 1);
        mfmas(0);
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            if (more) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 DS read
            __builtin_amdgcn_sched_group_barrier(0x008, NM / NL, 0);      // NM/NL MFMA
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
        stage_store<A_KC, TILE, WAVES>(An, tid, ga[nxt]);
        stage_store<B_KC, TILE, WAVES>(An + G::OPBUF, tid, gb[nxt]);
#pragma unroll
        for (int g = 0; g < NL; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM / NL, 0);
            if (more) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 DS write
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (more) read_frags(An, An + G::OPBUF, 0, 0);
        mfmas(1);
#pragma unroll
        for (int g = 0; g < NR / 2; ++g) {
            if (more) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM / (NR / 2), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        int t = 0;
        for (; t + RS <= nk; t += RS) {
            slab(t, std::integral_constant<int, 0>{});
            if (RS > 1) slab(t + 1, std::integral_constant<int, 1 % RS>{});
            if (RS > 2) slab(t + 2, std::integral_constant<int, 2 % RS>{});
        }
        if (RS > 1 && t < nk) slab(t, std::integral_constant<int, 0>{});
        if (RS > 2 && t + 1 < nk) slab(t + 1, std::integral_constant<int, 1 % RS>{});
    }

    // epilogue: acc[mi][ni][r] = C(i = i0+wm*WTM+mi*16+(lane&15), j = j0+wn*WTN+ni*16+(lane>>4)+4r)
    const bool mirror = a.mirror && (it != jt);
    double* cbase = a.C + (size_t)(j0 + wn * G::WTN + (lane >> 4)) * a.ldc + i0 + wm * G::WTM + (lane & 15);
    if (a.beta != 0.0) {
        // C tile read in batches of 8 independent loads before it is combined (not one load-wait per element)
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
#pragma unroll
            for (int n2 = 0; n2 < G::NI; n2 += 2) {
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
            for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = a.alpha * acc[mi][ni][r];
    }
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[mi][ni][r];
                cbase[(size_t)(ni * 16 + 4 * r) * a.ldc + mi * 16] = v;
                if (mirror) {
                    int i = i0 + wm * G::WTM + mi * 16 + (lane & 15);
                    int j = j0 + wn * G::WTN + ni * 16 + (lane >> 4) + 4 * r;
                    a.C[(size_t)i * a.ldc + j] = v;
                }
            }
        }
    }
}

template <bool A_KC, bool B_KC, int TILE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 2) void gemm_f64_kernel(GemmArgs a) {   // 2nd argument: waves per SIMD (2 workgroups per CU)
    using G = Geo<TILE, WAVES>;
    __shared__ __attribute__((aligned(16))) double lds[4 * G::OPBUF];
    // tile order table built on the host (tile_order.hip): workgroup b runs on XCD b%8,
    // every XCD walks its own work-balanced list of 2-D super-tiles (L2 locality).
    const uint32_t packed = a.order[blockIdx.x];
    if (packed == 0xffffffffu) return;
    gemm_tile_body<A_KC, B_KC, TILE, WAVES>(a, (int)(packed >> 16), (int)(packed & 0xffffu), lds);
}

// ---- fused small launches (la_kernels.h) -------------------------------------------------------------------------------------
// Device-wide barrier between two products of a fused launch.  Every workgroup arrives once per barrier; the counter never
// goes back, so barrier b of a launch that started at `base` is passed when the counter reaches base + (b + 1) * workgroups.
// Release / acquire at agent scope: the tiles a workgroup wrote are visible to every XCD's L2 before it arrives, and nothing
// it reads afterwards comes from a stale line.
__device__ __forceinline__ bool fused_grid_barrier(unsigned long long* counter, unsigned long long target, int* info) {
    __shared__ int timed_out;
    __syncthreads();
    if (threadIdx.x == 0) {
        timed_out = 0;
        __atomic_thread_fence(__ATOMIC_RELEASE);                    // (agent scope is the default of the HIP fence builtins below)
        __threadfence();
        atomicAdd(counter, 1ULL);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1L << 24)) {                              // ~2 s: another workgroup never arrived
                atomicMin(info, INFO_BARRIER_TIMEOUT);
                timed_out = 1;
                break;
            }
        }
        __threadfence();
    }
    __syncthreads();
    __threadfence();                                                // every wave: its later loads must miss the old lines
    return timed_out == 0;
}

__global__ __launch_bounds__(256, 2) void gemm_f64_fused_kernel(FusedArgs f) {
    using G = Geo<64, 4>;
    __shared__ __attribute__((aligned(16))) double lds[4 * G::OPBUF];
    const int nwg = (int)gridDim.x;
    for (int o = 0; o < f.nops; ++o) {
        const FusedOp& a = f.op[o];
        const int mt = 2 * a.mt, nt = 2 * a.nt;
        const int total = a.lower ? mt * (mt + 1) / 2 : mt * nt;
        for (int t = (int)blockIdx.x; t < total; t += nwg) {
            int it, jt;
            if (a.lower) {
                it = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while ((it + 1) * (it + 2) / 2 <= t) ++it;
                while (it * (it + 1) / 2 > t) --it;
                jt = t - it * (it + 1) / 2;
            } else {
                it = t % mt;
                jt = t / mt;
            }
            __syncthreads();        // the previous tile's last fragment reads
            if (!a.akc && !a.bkc)
                gemm_tile_body<false, false, 64, 4>(a, it, jt, lds);
            else if (!a.akc && a.bkc)
                gemm_tile_body<false, true, 64, 4>(a, it, jt, lds);
            else if (a.akc && a.bkc)
                gemm_tile_body<true, true, 64, 4>(a, it, jt, lds);
            else
                gemm_tile_body<true, false, 64, 4>(a, it, jt, lds);
        }
        if (o + 1 < f.nops && !fused_grid_barrier(f.counter, f.base + (unsigned long long)(o + 1) * (unsigned long long)nwg, f.info)) return;
    }
}

void launch_gemm_fused(const FusedArgs& f, int grid, hipStream_t s) {
    if (f.nops <= 0 || grid <= 0) return;
    hipLaunchKernelGGL(gemm_f64_fused_kernel, dim3(grid), dim3(256), 0, s, f);
}

// The throughput kernel: TILE = 128, operands staged with LDS-DMA (no staging registers, no ds_write), two separate
// LDS buffers (distinct __shared__ objects, so that the compiler knows a DMA into one never aliases the fragment
// reads of the other and does not serialise them behind vmcnt).
template <bool A_KC, bool B_KC, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 2) void gemm_f64_dma_kernel(GemmArgs a) {
    constexpr int TILE = 128;
    using G = Geo<TILE, WAVES>;
    __shared__ __attribute__((aligned(16))) double lds0[2 * G::OPBUF];
    __shared__ __attribute__((aligned(16))) double lds1[2 * G::OPBUF];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    uint32_t offa[G::NQ], offb[G::NQ];
    dma_offsets<A_KC, TILE, WAVES>(a.lda, wave, lane, offa);
    dma_offsets<B_KC, TILE, WAVES>(a.ldb, wave, lane, offb);
    // One table entry per workgroup, or two (a.pairs): a workgroup then computes two tiles one after the other, the second one
    // walked TOWARDS the k all tiles have in common (bit 15 of the entry), see tile_order.hip.
    const int nparts = a.pairs ? 2 : 1;
#pragma nounroll
    for (int part = 0; part < nparts; ++part) {
    const uint32_t packed = a.order[(size_t)blockIdx.x * nparts + part];
    if (packed == 0xffffffffu) break;
    const int it = (int)(packed >> 16), jt = (int)(packed & 0x7fffu);
    const bool flip = (packed & 0x8000u) != 0;
    int kbeg = 0, kend = a.K;
    switch (a.kmode) {
        case KM_LE_J: kend = (jt + 1) * 128; break;
        case KM_GE_J: kbeg = jt * 128; break;
        case KM_LE_I: kend = (it + 1) * 128; break;
        case KM_GE_I: kbeg = it * 128; break;
        default: break;
    }
    if (kend > a.K) kend = a.K;
    const int i0 = it * TILE, j0 = jt * TILE;
    if (part) __syncthreads();       // the first tile's last fragment reads are done before the next DMA lands

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
    const bool down = ((a.kmode == KM_GE_J || a.kmode == KM_GE_I) && !a.k_ascending) != flip;
    auto issue = [&](int t, double* buf) {
        const int k0 = down ? kend - (t + 1) * 16 : kbeg + t * 16;
        dma_issue<A_KC, TILE, WAVES>(stage_base<A_KC>(a.A, a.lda, i0, k0), offa, buf, wave);
        dma_issue<B_KC, TILE, WAVES>(stage_base<B_KC>(a.B, a.ldb, j0, k0), offb, buf + G::OPBUF, wave);
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
    const bool mirror = a.mirror && (it != jt);
    double* cbase = a.C + (size_t)(j0 + wn * G::WTN + (lane >> 4)) * a.ldc + i0 + wm * G::WTM + (lane & 15);
    if (a.beta != 0.0) {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
#pragma unroll
            for (int n2 = 0; n2 < G::NI; n2 += 2) {
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
            for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = a.alpha * acc[mi][ni][r];
    }
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = acc[mi][ni][r];
                cbase[(size_t)(ni * 16 + 4 * r) * a.ldc + mi * 16] = v;
                if (mirror) {
                    int i = i0 + wm * G::WTM + mi * 16 + (lane & 15);
                    int j = j0 + wn * G::WTN + ni * 16 + (lane >> 4) + 4 * r;
                    a.C[(size_t)i * a.ldc + j] = v;
                }
            }
        }
    }
    }   // part
}

template <int WAVES>
static void launch_gemm_dma(const GemmArgs& a, int a_kc, int b_kc, hipStream_t s) {
    dim3 grid(a.grid), block(64 * WAVES);
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
    dim3 grid(a.grid), block(64 * WAVES);
    if (!a_kc && !b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<false, false, TILE, WAVES>), grid, block, 0, s, a);
    else if (!a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<false, true, TILE, WAVES>), grid, block, 0, s, a);
    else if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f64_kernel<true, true, TILE, WAVES>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((gemm_f64_kernel<true, false, TILE, WAVES>), grid, block, 0, s, a);
}

// DNAGPU_GEMM_VARIANT selects the 128-tile kernel for A/B comparisons: "dma4" (default), "dma8", "reg8", "reg4"
static int gemm_variant_128() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DNAGPU_GEMM_VARIANT");
        v = 1;
        if (e && !strcmp(e, "dma8")) v = 0;
        if (e && !strcmp(e, "reg8")) v = 2;
        if (e && !strcmp(e, "reg4")) v = 3;
    }
    return v;
}

bool gemm_128_takes_pairs() { return gemm_variant_128() <= 1; }

void launch_gemm(const GemmArgs& a, int a_kc, int b_kc, hipStream_t s) {
    if (a.grid <= 0) return;
    if (a.tile == 64) {
        launch_gemm_t<64, 4>(a, a_kc, b_kc, s);
        return;
    }
    static const int k_asc = getenv("DNAGPU_K_ASCENDING") ? 1 : 0;   // diagnostic A/B switch
    GemmArgs b = a;
    b.k_ascending = k_asc;
    switch (gemm_variant_128()) {
        case 1: launch_gemm_dma<4>(b, a_kc, b_kc, s); break;
        case 2: launch_gemm_t<128, 8>(b, a_kc, b_kc, s); break;
        case 3: launch_gemm_t<128, 4>(b, a_kc, b_kc, s); break;
        default: launch_gemm_dma<8>(b, a_kc, b_kc, s); break;
    }
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
