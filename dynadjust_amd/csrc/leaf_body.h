// The leaf of the recursive inverse as a device function: potrf + trtri of one 128 x 128 diagonal tile by ONE workgroup of 8 waves
// (leaf_kernel.hip).  The work is dealt to the waves by 16 x 16 block.
// Replaces dpotrf / dtrtri on the diagonal blocks of matrix_2d::cholesky_inverse (dnamatrix_contiguous.cpp:982-1006).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnagpu {

#ifdef DNAGPU_LEAF_PROBE
__device__ unsigned long long leaf_probe[64];
#define LEAF_PROBE(i) do { if (threadIdx.x == 0) leaf_probe[i] = __builtin_readcyclecounter(); } while (0)
#define LEAF_PROBE_W1(i) do { if (threadIdx.x == 64) leaf_probe[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LEAF_PROBE(i) do { } while (0)
#define LEAF_PROBE_W1(i) do { } while (0)
#endif

namespace leaf {
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// acc += sign * A * B for 16x16 operands in LDS: A(i,k) = a[i*ars + k*acs], B(k,j) = b[k*brs + j*bcs].
// v_mfma_f64_16x16x4_f64: lane l feeds A(l&15, l>>4) and B(l>>4, l&15); acc[r] = D((l>>4) + 4r, l&15).
__device__ __forceinline__ d4 mma16(const double* a, int ars, int acs, const double* b, int brs, int bcs, d4 acc, double sign, int lane) {
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * kk + hi;
        double av = sign * a[lo * ars + k * acs];
        double bv = b[k * brs + lo * bcs];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    return acc;
}


// The tile lives in LDS as its 36 lower 16x16 blocks (block (bi, bj), bi >= bj, number bi (bi + 1) / 2 + bj, rows BR apart):
// 76.5 KiB instead of 149 KiB for the square tile + separate diagonal inverses, so that a leaf can share a CU with one workgroup
// of the tile GEMM (72 KiB).  `blk(bi, bj)` of the caller says where a block lives (one array in the leaf kernel).
constexpr int BR = 17;        // row stride inside a block (odd: conflict-free column reads)
constexpr int BS = 16 * BR;   // doubles per block

__device__ __forceinline__ d4 tile_load(const double* B, int lane) {
    const int lo = lane & 15, hi = lane >> 4;
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = B[(hi + 4 * r) * BR + lo];
    return v;
}

__device__ __forceinline__ void tile_store(double* B, int lane, d4 v) {
    const int lo = lane & 15, hi = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) B[(hi + 4 * r) * BR + lo] = v[r];
}

// Wave 0 only: the 16 x 16 diagonal block xd (LDS, rows BR apart) factored in registers; what is left in xd is the INVERSE of its
// factor (zeros above the diagonal).  col0: the block's first column in the matrix (info).
__device__ __forceinline__ void diag_block(double* xd, double* const LT, int lane, int* info, int col0) {
    const int i = lane & 15;
    const bool is_d = lane < 16;
    int bad = 16;
    // Right-looking, one column per step.  The pivot chain (pivot -> rsqrt -> scaled column -> next pivot) only needs the
    // updates of the next two columns at once: those two multipliers come by v_readlane; the others travel through LDS
    // (column k written by its lanes, read back as broadcasts) and are applied one step later, after the next pivot's
    // Newton iterations.  Every element still receives its updates in ascending k: same bits as one column at a time.
    // The same multipliers drive the forward substitution D X = I (lane j solves column j:
    // x_i = (delta_ij - sum_{k<i} L_ik x_k) / L_ii) in the same steps -- as v_readlane values in a loop of its own they
    // were kept in 240 SGPRs and spilled: the diagonal blocks were 57 % of the leaf.
    // Factor and substitution share their instructions: u[] is row i of the block in lanes 0..15 and column i of X in
    // the other lanes (three copies), and both obey u[j] -= L(j,k) u[k], u[k] *= 1 / L(k,k).
    double u[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) u[j] = is_d ? xd[i * BR + j] : (j == i ? 1.0 : 0.0);
    double lp[16];                      // column k - 1 of L, rows k + 2 .. 15 (in flight during step k)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double pk = readlane_f64(u[k], k);
        if (!(pk > 0.0)) {          // (uniform: a scalar select; reported once per block, below)
            bad = bad < k ? bad : k;
            pk = 1.0;
        }
        // y = pk^-1/2 (v_rsq_f64 + 2 Newton steps).  The diagonal element L(k,k) itself is never needed: the panel below
        // is solved with D^-1, and every later step reads the column's rows below k only
        double y = __builtin_amdgcn_rsq(pk);
        const double h = 0.5 * pk;
        y = y * fma(-h * y, y, 1.5);
        y = y * fma(-h * y, y, 1.5);
        __builtin_amdgcn_sched_barrier(0);   // (a late update hoisted above the Newton steps would wait for LDS there)
        // the late updates of column k - 1 (rows k + 2 ..; rows k, k + 1 were done in step k - 1)
        if (k > 0) {
#pragma unroll
            for (int j = k + 2; j < 16; ++j) {
                u[j] = fma(-lp[j], u[k - 1], u[j]);
                asm volatile("" : "+v"(u[j]));                   // (here, not sunk to the store below with lp[] kept alive)
            }
        }
        u[k] *= y;
        if (is_d) LT[k * 16 + i] = u[k];                         // LT(k, i) = L(i, k)
#pragma unroll
        for (int j = k + 1; j < 16 && j <= k + 2; ++j) {
            double ljk = readlane_f64(u[k], j);
            u[j] = fma(-ljk, u[k], u[j]);
            asm volatile("" : "+v"(u[j]));
        }
        asm volatile("" ::: "memory");                           // (LDS is in order within a wave: no wait needed)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = k + 3; j < 16; ++j) lp[j] = LT[k * 16 + j];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (bad < 16 && lane == 0) atomicMin(info, col0 + bad + 1);
    if (!is_d) {
#pragma unroll
        for (int ii = 0; ii < 16; ++ii) xd[ii * BR + i] = u[ii];  // X(ii, i); zero above the diagonal
    }
}

// ---- the overlapped schedule (round 4) ----
// The same operations on every 16 x 16 block in the same order as the first, lock-step version (eight phases per 16-column panel, removed in
// round 5) -- same bits -- but the serial part, the diagonal blocks of wave 0 (8 x 4 600 of the old leaf's 91 600 clocks), no longer has
// the other waves wait for it:
//   * phase B's step kb (X = L^-1, column block by column block) only touches block columns <= kb, phase A's steps > kb only block
//     columns > kb: step kb of phase B runs beside the diagonal block kb + 1, and so does the trailing update of step kb but for the
//     one block wave 0 needs (which it updates itself);
//   * inside phase B's step a block column belongs to one wave (LDS is in order within a wave): no barrier of its own; the step's last
//     operation (the panel's own column) moves to the next panel's round;
//   * the tile arrives in two parts (block column 0, then the rest beside diagonal block 0) and X leaves row block by row block
//     as the rows become final.
// One barrier per panel and a counter in LDS; 38 -> 29 us.
struct TileOp {
    double* C;          // C <- (use_c ? C : 0) + sign * A * B'
    const double* A;
    const double* B;    // B'(k, j) = bt ? B[j * BR + k] : B[k * BR + j]
    int bt, use_c;
    double sign;
    int dep;            // reads what the wave's previous operation writes: its operands are not fetched ahead
};
struct TileRegs {
    d4 c;
    double a[4], b[4];
};
__device__ __forceinline__ void op_fetch(const TileOp& op, TileRegs& r, int lane) {
    const int lo = lane & 15, hi = lane >> 4;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    r.c = op.use_c ? tile_load(op.C, lane) : zero;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        r.a[kk] = op.A[lo * BR + 4 * kk + hi];           // (the sign is applied in op_run: nothing here waits for a load)
        r.b[kk] = op.bt ? op.B[lo * BR + 4 * kk + hi] : op.B[(4 * kk + hi) * BR + lo];
    }
}
__device__ __forceinline__ void op_run(const TileOp& op, TileRegs& r, int lane) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) r.c = __builtin_amdgcn_mfma_f64_16x16x4f64(op.sign * r.a[kk], r.b[kk], r.c, 0, 0, 0);
    tile_store(op.C, lane, r.c);
    asm volatile("" ::: "memory");
}
// n operations of one wave, op(idx) = gen(idx): the operands of the next one on their way from LDS during the MFMAs of the current one
template <class Gen>
__device__ __forceinline__ void run_ops(int n, int lane, Gen gen) {
    if (n <= 0) return;
    TileOp cur = gen(0);
    TileRegs rc;
    op_fetch(cur, rc, lane);
    for (int idx = 0; idx < n; ++idx) {
        const bool more = idx + 1 < n;
        TileOp nxt = cur;
        TileRegs rn = rc;
        if (more) {
            nxt = gen(idx + 1);
            if (!nxt.dep) op_fetch(nxt, rn, lane);
        }
        op_run(cur, rc, lane);
        if (more && nxt.dep) op_fetch(nxt, rn, lane);
        cur = nxt;
        rc = rn;
    }
}

// meet: two ints of LDS -- [0] the counter the waves other than wave 0 meet at, [1] the last 16-column block of the tile that holds
// anything but identity padding (round 5): a tile whose trailing blocks are padding -- the ragged end of every matrix, the whole second
// tile of a 150-unknown junction -- stops after its real blocks; every step it leaves out would have multiplied zeros and ones
// (the padding's factor and inverse are the identity, its panels zero), so the results are the same bits.
template <int NW, class Blk>
__device__ __forceinline__ void potrf_trtri_tile_overlapped(const double* __restrict__ A, int lda, double* __restrict__ X, int ldx, int o, int* info,
                                                            double* const LT, int* const meet, Blk blk) {
    static_assert(NW == 8, "one block column of phase B per wave 1..7");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NO = (NW - 1) * 64;     // threads of the waves other than wave 0
    const int oid = tid - 64;             // ... numbered

    LEAF_PROBE(0);
    if (tid == 0) {
        meet[0] = 0;
        meet[1] = 0;
    }
    int real_blk = 0;       // the last row block this thread saw something other than padding in (an entry off the diagonal, a diagonal entry != 1)
    // block column 0 (16 columns x 128 rows: 4 elements per thread)
    {
        const int row = tid & 127, q = tid >> 7;
        double v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = q + 4 * t;
            v[t] = (row >= c) ? A[(size_t)c * lda + row] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = q + 4 * t;
            blk(row >> 4, 0)[(row & 15) * BR + c] = v[t];
            if (row >= c && !(v[t] == (row == c ? 1.0 : 0.0))) real_blk = max(real_blk, row >> 4);
        }
    }
    __syncthreads();
    LEAF_PROBE(1);
    if (wave == 0) {
        if (real_blk) atomicMax(&meet[1], real_blk);
        __builtin_amdgcn_s_setprio(2);
        diag_block(blk(0, 0), LT, lane, info, o);
        __builtin_amdgcn_s_setprio(0);
    } else {
        // the other 112 columns: 32 elements per thread, all loads in flight together
        double v[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int e = oid + NO * t, row = e & 127, c = 16 + (e >> 7);
            v[t] = (row >= c) ? A[(size_t)c * lda + row] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int e = oid + NO * t, row = e & 127, c = 16 + (e >> 7);
            if ((row >> 4) >= (c >> 4)) blk(row >> 4, c >> 4)[(row & 15) * BR + (c & 15)] = v[t];
            if (row >= c && !(v[t] == (row == c ? 1.0 : 0.0))) real_blk = max(real_blk, row >> 4);
        }
        if (real_blk) atomicMax(&meet[1], real_blk);
    }
    __syncthreads();
    LEAF_PROBE(2);
    const int nreal = __builtin_amdgcn_readfirstlane(meet[1]) + 1;      // blocks 0 .. nreal - 1 hold the matrix, the rest is identity padding

#pragma unroll 1
    for (int kb = 0; kb < nreal; ++kb) {
        const double* xd = blk(kb, kb);
        // round P: the panel below diagonal block kb, P = A_panel * D^-T (7 - kb blocks), and -- phase B, last operation of step
        // kb - 1, after every reader of L(i, kb-1) -- M(i, kb-1) = -L(i, kb-1) D_(kb-1)^-1, i >= kb (8 - kb blocks): all independent.
        // Wave 0 computes the panel block under the next diagonal block, signals, and goes on without waiting for the others; they
        // share the rest and meet at a counter in LDS (the hardware barrier has no arrive-without-wait on this part).
        const int np = 7 - kb, n3 = kb > 0 ? 8 - kb : 0;
        if (wave == 0) {
            if (kb < 7) {
                __builtin_amdgcn_s_setprio(2);
                TileOp op;
                op.C = blk(kb + 1, kb); op.A = op.C; op.B = xd; op.bt = 1; op.use_c = 0; op.sign = 1.0; op.dep = 0;
                TileRegs r;
                op_fetch(op, r, lane);
                op_run(op, r, lane);
            }
            if (lane == 0) __hip_atomic_fetch_add(meet, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (kb < 7) {
                // the one block of the trailing update that the next diagonal block needs, then that block
                TileOp op;
                op.C = blk(kb + 1, kb + 1); op.A = blk(kb + 1, kb); op.B = op.A; op.bt = 1; op.use_c = 1; op.sign = -1.0; op.dep = 0;
                TileRegs r;
                op_fetch(op, r, lane);
                op_run(op, r, lane);
                diag_block(blk(kb + 1, kb + 1), LT, lane, info, o + 16 * (kb + 1));
                __builtin_amdgcn_s_setprio(0);
            }
        } else {
            const int w = wave - 1;
            {
                const int first = kb < 7 ? 1 : 0;          // (operation 0 of the panel is wave 0's)
                const int n = np + n3 - first;
                const int mine = n > w ? (n - w + NW - 2) / (NW - 1) : 0;
                run_ops(mine, lane, [&](int idx) {
                    const int q = first + w + (NW - 1) * idx;
                    TileOp op;
                    if (q < np) {
                        op.C = blk(kb + 1 + q, kb); op.A = op.C; op.B = xd; op.bt = 1; op.use_c = 0; op.sign = 1.0; op.dep = 0;
                    } else {
                        op.C = blk(kb + (q - np), kb - 1); op.A = op.C; op.B = blk(kb - 1, kb - 1); op.bt = 0; op.use_c = 0; op.sign = -1.0; op.dep = 0;
                    }
                    return op;
                });
                if (lane == 0) __hip_atomic_fetch_add(meet, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(meet, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < NW * (kb + 1)) __builtin_amdgcn_s_sleep(1);
            }
            LEAF_PROBE_W1(3 + 2 * kb);
            if (w < kb) {
                // phase B, step kb, block column tj = w:  M(kb, tj) = D_kb^-1 M(kb, tj);  M(i, tj) -= L(i, kb) M(kb, tj), i > kb
                const int tj = w;
                run_ops(1 + (7 - kb), lane, [&](int idx) {
                    TileOp op;
                    if (idx == 0) {
                        op.C = blk(kb, tj); op.A = xd; op.B = op.C; op.bt = 0; op.use_c = 0; op.sign = 1.0; op.dep = 0;
                    } else {
                        const int i = kb + idx;
                        op.C = blk(i, tj); op.A = blk(i, kb); op.B = blk(kb, tj); op.bt = 0; op.use_c = 1; op.sign = -1.0; op.dep = (idx == 1);
                    }
                    return op;
                });
            } else {
                // the rest of the trailing update of step kb: tiles t = 1 .. T - 1 of the (7 - kb) x (7 - kb) lower triangle, round-robin
                // over the waves without a block column of phase B (7 - kb of them)
                const int nt = 7 - kb, T = nt * (nt + 1) / 2, v = w - kb;
                const int ntr = (T - 1 > v) ? (T - 1 - v + nt - 1) / nt : 0;
                run_ops(ntr, lane, [&](int idx) {
                    const int t = 1 + v + nt * idx;
                    int ti = 0;
                    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
                    const int tj = t - ti * (ti + 1) / 2;
                    TileOp op;
                    op.C = blk(kb + 1 + ti, kb + 1 + tj); op.A = blk(kb + 1 + ti, kb); op.B = blk(kb + 1 + tj, kb);
                    op.bt = 1; op.use_c = 1; op.sign = -1.0; op.dep = 0;
                    return op;
                });
            }
            // row block kb - 1 of X is final: out it goes (16 rows x 128 columns, zeros right of the diagonal)
            if (kb > 0) {
                const int rb = kb - 1;
                constexpr int NT = (2048 + NO - 1) / NO;
                double v[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int e = oid + NO * t, r = e & 15, c = (e >> 4) & 127, row = 16 * rb + r;
                    v[t] = (row >= c) ? blk(rb, c >> 4)[r * BR + (c & 15)] : 0.0;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int e = oid + NO * t, r = e & 15, c = e >> 4, row = 16 * rb + r;
                    if (e < 2048) X[(size_t)c * ldx + row] = v[t];
                }
            }
        }
        __syncthreads();
        LEAF_PROBE(4 + 2 * kb);
    }
    // the row blocks the loop has not written: the last one always; after a shortened loop also the last real one and the padding's
    // (identity rows, as they were loaded)
#pragma unroll 1
    for (int rb = nreal - 1; rb < 8; ++rb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = tid + 512 * t, r = e & 15, c = e >> 4, row = 16 * rb + r;
            const double v = (row >= c) ? blk(rb, c >> 4)[r * BR + (c & 15)] : 0.0;
            X[(size_t)c * ldx + row] = v;
        }
    }
    LEAF_PROBE(20);
}

}  // namespace leaf
}  // namespace dnagpu
