// The register-staged fp64 tile product (TILE = 64 / 4 waves x 32 x 32 and TILE = 32 / 4 waves x 16 x 16: the latency shapes of the small
// and tiny products of the recursion) as a device function of gemm_f64_kernel (la_kernels.hip).
#pragma once
#include "gemm_tile_dma.h"

namespace dnagpu {

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


// One TILE x TILE tile of C at (i0, j0), k over [kbeg, kend) ascending, by the calling workgroup of 64 * WAVES threads; operands
// staged through registers into the two LDS buffers ldsA / ldsB (2 * OPBUF doubles each).
// A: the launch's arguments (order / grid unused here).  The caller separates consecutive tiles of one workgroup by a barrier.
template <bool A_KC, bool B_KC, int TILE, int WAVES>
__device__ __forceinline__ void reg_tile_product(const double* A, int lda, const double* B, int ldb, double* C, int ldc, const int i0, const int j0,
                                                 const int kbeg, const int kend, double alpha, double beta, bool mirror, double* ldsA, double* ldsB) {
    using G = Geo<TILE, WAVES>;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

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
    constexpr int RS = (TILE <= 64) ? 3 : 1;
    d2 ga[RS][G::NQ], gb[RS][G::NQ];
    uint32_t offa[G::NQ], offb[G::NQ];
    stage_offsets<A_KC, TILE, WAVES>(lda, tid, offa);
    stage_offsets<B_KC, TILE, WAVES>(ldb, tid, offb);
    const int nk = (kend - kbeg) / 16;
    // beta != 0 on the small tiles: the C tile is on its way from the start (its 2 us of latency were the tail of a 6 us launch)
    constexpr bool CPRE = TILE <= 64;
    const double* cpre = C + (size_t)(j0 + wn * G::WTN + (lane >> 4)) * ldc + i0 + wm * G::WTM + (lane & 15);
    double cold0[CPRE ? G::MI : 1][CPRE ? G::NI : 1][4];
    if (CPRE && beta != 0.0) {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) cold0[mi][ni][r] = cpre[(size_t)(ni * 16 + 4 * r) * ldc + mi * 16];
    }
    // (slabs beyond the last one are clamped to it: loaded again, stored to the idle buffer, never used -- the slab body stays
    // branch free)
    auto load_slab = [&](int sl, auto slot_tag) {
        constexpr int slot = decltype(slot_tag)::value;
        const int k0 = kbeg + (sl < nk ? sl : nk - 1) * 16;
        stage_load<G::NQ>(stage_base<A_KC>(A, lda, i0, k0), offa, ga[slot]);
        stage_load<G::NQ>(stage_base<B_KC>(B, ldb, j0, k0), offb, gb[slot]);
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
        stage_store<A_KC, TILE, WAVES>(ldsA, tid, ga[0]);
        stage_store<B_KC, TILE, WAVES>(ldsA + G::OPBUF, tid, gb[0]);
    }
    __syncthreads();
    if (nk > 0) read_frags(ldsA, ldsA + G::OPBUF, 0, 0);

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
    constexpr bool HINTS = NM >= NL && NM >= NR / 2;   // (TILE = 32 has ONE MFMA per k-step: nothing to interleave, the compiler's order stands)
    auto slab = [&](int t, auto slot_tag) {
        constexpr bool more = true;
        constexpr int slot = decltype(slot_tag)::value;      // t % RS: free since slab t went to LDS; slab t + RS moves in
        constexpr int nxt = (slot + 1) % RS;                 // slab t + 1: goes to LDS during this slab
        const int cur = t & 1;
        const double* As = cur ? ldsB : ldsA;
        const double* Bs = As + G::OPBUF;
        double* An = cur ? ldsA : ldsB;
        load_slab(t + RS, slot_tag);
        read_frags(As, Bs, 1, 1);
        mfmas(0);
        if constexpr (HINTS) {
#pragma unroll
            for (int g = 0; g < NL; ++g) {
                if (more) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, NM / NL, 0);      // NM/NL MFMA
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frags(As, Bs, 2, 0);
        mfmas(1);
        if constexpr (HINTS) {
#pragma unroll
            for (int g = 0; g < NR / 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NM / (NR / 2), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
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
    double* cbase = C + (size_t)(j0 + wn * G::WTN + (lane >> 4)) * ldc + i0 + wm * G::WTM + (lane & 15);
    if (CPRE && beta != 0.0) {
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mi][ni][r] = alpha * acc[mi][ni][r] + beta * cold0[mi][ni][r];
    } else if (beta != 0.0) {
        // C tile read in batches of 8 independent loads before it is combined (not one load-wait per element)
#pragma unroll
        for (int mi = 0; mi < G::MI; ++mi) {
            constexpr int NB = G::NI >= 2 ? 2 : 1;       // (TILE = 32: one MFMA tile per wave along j)
#pragma unroll
            for (int n2 = 0; n2 < G::NI; n2 += NB) {
                double cold[NB][4];
#pragma unroll
                for (int ni = 0; ni < NB; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cold[ni][r] = cbase[(size_t)((n2 + ni) * 16 + 4 * r) * ldc + mi * 16];
#pragma unroll
                for (int ni = 0; ni < NB; ++ni)
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
