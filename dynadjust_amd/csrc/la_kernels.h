// Internal (C++) interface of the dense fp64 kernels.  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

namespace dnagpu {

enum KMode { KM_FULL = 0, KM_LE_J = 1, KM_GE_J = 2, KM_LE_I = 3, KM_GE_I = 4 };

// A batched launch: the same product (or leaf) on `nb` matrices of one shape, member b's operands at the member-0 addresses plus
// an element offset of their own (their buffers are separate allocations).  The workgroups of member 1 follow those of member 0
// in the same launch: no launch boundary, no partly filled last wave between them, and the short launches at the bottom of the
// recursion have nb times the tiles (sym_inverse.h: InvBatch).
constexpr int BATCH_MAX = 32;      // (members of a merged launch: 32 for the steps of a chain plan, include/dnagpu.h DNAGPU_CHAIN_BATCH_MAX; block batches stay at 16)

struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    int lda, ldb, ldc;
    int mt, nt;  // 128-wide tiles in M and N (the kernel re-tiles them when tile == 64)
    int K;       // multiple of 16
    double alpha, beta;
    int kmode;   // KMode: restricts the k range per tile (triangular operands)
    int lower;   // 1: only tiles it >= jt (square problems)
    int mirror;  // 1: also store C(j,i) for off-diagonal tiles
    const uint32_t* order;  // device table: (it << 16 | jt) per workgroup, 0xffffffff = idle
    int grid;               // number of workgroups (= table length)
    int tile;               // block tile of the launch: 128 (throughput), 64 (small launches) or 32 (tiny ones)
    int nb = 1;             // members of a batched launch (grid.y); member b works on A + dA[b], B + dB[b], C + dC[b]
    long long dA[BATCH_MAX], dB[BATCH_MAX], dC[BATCH_MAX];
};

struct LeafBatch {
    int nb = 1;
    long long dA[BATCH_MAX], dX[BATCH_MAX];     // member b: A + dA[b], X + dX[b], info[b]
};

// launches with fewer 128-tiles than this run on 64x64 block tiles
// (cfg3, r02: 160 / 384 / 768 -> 3.83 / 3.79 / 3.77 s per step with four chains, 4.38 / 4.33 / 4.35 s with one)
constexpr int SMALL_LAUNCH_TILES = 512;
// ... and with fewer 128-tiles than this on 32 x 32 block tiles (round 4).  A 64-tile of K = 512 is 13.7 us of MFMA time on
// the ONE CU it occupies, and a product of a dozen 128-tiles occupies a fifth of the chip: sixteen times the workgroups of the 128-tile shape
// spread the same flops over every CU there is.  Same bits (an element's k order does not depend on the tile it is computed in).
// Measured (profiles/r04_tiny_tiles.txt; thresholds 0 / 8 / 16 / 32 / 64 / 128 / 256 / 512): inverse n = 6 144 27.7 -> 31.5 TFLOP/s, elimination
// n = 19 968 49.7 -> 52.7, cfg2 404 -> 397 ms, cfg3 one chain 2 626 -> 2 575 ms, the small-block workload 569 -> 516 ms; 64 is where the
// batched workloads stop gaining (single matrices gain another 1 - 2 % up to 256).
constexpr int TINY_LAUNCH_TILES = 64;

// Build the workgroup -> tile table for a launch shape (host side, see tile_order.cpp).
// Returns the table (length = grid, multiple of 8 when more than 8 tiles).
// jt_lo / jt_hi (128-tiles, -1 = all): only the tiles of these columns (one rank's share of a split launch)
std::vector<uint32_t> build_tile_order(int mt, int nt, int K, int kmode, int lower, int tile, int jt_lo = -1, int jt_hi = -1);
std::vector<int> split_tile_columns(int mt, int nt, int K, int kmode, int lower, int world);

void launch_gemm(const GemmArgs& a, int a_kcontig, int b_kcontig, hipStream_t s);
void launch_leaf(const double* A, int lda, double* X, int ldx, int o, int* info, hipStream_t s, const LeafBatch* batch = nullptr);
void launch_unpack_lower(const double* ap, double* F, uint32_t n, uint32_t np, hipStream_t s);
void launch_pack_lower(const double* F, double* ap, uint32_t n, uint32_t np, hipStream_t s);
void launch_init_padded(double* F, uint32_t n, uint32_t np, hipStream_t s);
void launch_init_padding(double* F, uint32_t n, uint32_t np, hipStream_t s);   // rows / columns n .. np - 1 only
void launch_diag_rsqrt(const double* F, double* s, uint32_t n, uint32_t np, hipStream_t st);
void launch_scale_sym(double* F, const double* s, uint32_t n, uint32_t np, int lower_only, hipStream_t st);
void launch_symmetrize(double* F, uint32_t n, uint32_t np, hipStream_t s);
void launch_symv(const double* F, const double* x, double* y, double* part, uint32_t n, uint32_t np, uint32_t nchunks,
                 hipStream_t st);

inline uint32_t pad128(uint32_t n) { return n == 0 ? 128u : ((n + 127u) / 128u) * 128u; }

// DNAGPU_POISON_ALLOC=1 (diagnostic): every device allocation of the library is filled with 0xFF bytes -- NaN as a double -- so that a
// buffer that is read before it was written shows up as NaN in the results instead of as whatever the memory held before (typically
// the same values from an earlier adjustment of the process: invisible)
inline hipError_t poison_malloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    static const bool poison = getenv("DNAGPU_POISON_ALLOC") && atoi(getenv("DNAGPU_POISON_ALLOC")) != 0;
    if (e == hipSuccess && poison && bytes) {
        e = hipMemset(*p, 0xFF, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    return e;
}
template <class T>
inline hipError_t poison_malloc(T** p, size_t bytes) { return poison_malloc(reinterpret_cast<void**>(p), bytes); }


}  // namespace dnagpu
