// Leaf of the recursive inverse: one 128x128 diagonal tile, one workgroup (8 waves).
// Reads the lower triangle of A(o:o+128, o:o+128), factors it (L L^T) and writes X = L^-1
// (lower, zeros above the diagonal) to the X buffer.  The tile lives in LDS as its lower 16x16 blocks (78.5 KiB of the
// CU's 160 KiB: room for a tile-GEMM workgroup beside it).  Both phases are blocked by 16-column panels
// (leaf_body.h; the operations on every 16 x 16 block and their order are those of the first, lock-step version -- same bits):
//   phase A (Cholesky), per panel kb
//     [wave 0]   factor the 16x16 diagonal block D and invert it: one row / column per lane in registers, v_rsq_f64 +
//                Newton, the two multipliers the pivot chain needs at once by v_readlane, the others as LDS broadcasts
//     [waves]    P = A_panel * D^-T            one 16 x 16 block per wave, 4 MFMA f64 16x16x4
//     [waves]    A_trail -= P P^T              16x16 tiles round-robin over the waves
//   phase B (X = L^-1, in place), per panel kb -- the D^-1 of phase A are reused
//     [waves]    M(kb, :kb) = D^-1 * M(kb, :kb)
//     [waves]    M(i, :kb) -= L(i,kb) * M(kb, :kb)     a block column per wave
//     [waves]    M(i,kb) = -L(i,kb) * D^-1
// Round 4: wave 0 runs the serial chain (diagonal block, the panel block below it, that block's update of the next diagonal block,
// the next diagonal block ...) without ever waiting for the others; beside diagonal block kb + 1 the other seven waves do the rest
// of step kb's trailing update AND step kb of phase B (which touches block columns <= kb only), read the tile's other 112 columns
// in (beside diagonal block 0) and write X's finished row blocks out.  One hardware barrier per panel (+ a counter in LDS the
// seven waves meet at): 38 -> 29 us (tools/leaf_probe.hip).  A non-positive or NaN pivot records
// info = global column + 1 (dpotrf's info; the facade turns it into the reference's
// "Matrix inversion failed, the matrix is singular.", dnamatrix_contiguous.cpp:983).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "la_kernels.h"
#include "leaf_body.h"

namespace dnagpu {

__global__ __launch_bounds__(512) void leaf_potrf_trtri_kernel(const double* __restrict__ A, int lda, double* __restrict__ X,
                                                               int ldx, int o, int* info, LeafBatch batch) {
    // LT: the factor of the current diagonal block, transposed (wave 0 only) -- first, so that its constant addresses fit the
    // 16-bit offset field of the ds instructions (behind S they took a register each)
    __shared__ double SH[256 + 36 * leaf::BS];
    __shared__ int meet[2];
    double* const S = SH + 256;
    if (batch.nb > 1) {     // one workgroup per member of the batch
        const int b = (int)blockIdx.x;
        A += batch.dA[b];
        X += batch.dX[b];
        info += b;
    }
    leaf::potrf_trtri_tile_overlapped<8>(A + (size_t)o * lda + o, lda, X + (size_t)o * ldx + o, ldx, o, info, SH, meet, [S](int bi, int bj) { return S + (bi * (bi + 1) / 2 + bj) * leaf::BS; });
}

void launch_leaf(const double* A, int lda, double* X, int ldx, int o, int* info, hipStream_t s, const LeafBatch* batch) {
    LeafBatch one;
    const LeafBatch& b = batch ? *batch : one;
    hipLaunchKernelGGL(leaf_potrf_trtri_kernel, dim3(b.nb), dim3(512), 0, s, A, lda, X, ldx, o, info, b);
}

}  // namespace dnagpu
