"""Full-size parity records: a fixed sample of every block's result (all coordinates, the variance matrix's diagonal and a
few of its columns) that is small enough to commit, taken the same way from the CPU oracle and from the device path.
Used by tools/make_fullsize_golden.py (writes tests/golden/<workload>_oracle.npz on a host with enough memory for the
oracle) and by tests/test_gpu_fullsize.py (compares the device path with it at BASELINE.json's full sizes)."""
import numpy as np

# name: (rows, cols, baselines, blocks, phased) -- the bench's workloads (bench.py WORKLOADS)
WORKLOADS = {
    "cfg3": (316, 317, 266666, 16, True),
    "cfg2": (100, 100, 26666, 1, False),
    # a quarter of cfg3: 4 of its 16 strips at the same block size (n ~ 20 000, 317-station junction rows) -- the oracle's run of
    # it fits a 64 GB host (cfg3 itself needs ~80 GB and 7.4e14 flops on the CPU)
    "cfg3q": (79, 317, 66666, 4, True),
    # four of cfg4's 128 strips at cfg4's block geometry: n ~ 27 000 unknowns per block, junction rows of 1 000 stations (J = 3 000),
    # condensed blocks of 6 000 unknowns -- 20 Solve() calls, ~4e14 flops and ~45 GB on the CPU
    "cfg4q": (32, 1000, 85000, 4, True),
    # bench.py's `smallblocks`: the same 100 000 stations cut the way dnasegment's defaults would -- strips of 1 ... 10 rows of 150 stations,
    # 120 blocks of n = 900 ... 4 950 -- whole: 1 074 Solve() calls, 4.4e13 flops, a minute on the CPU
    "smallblocks": (668, 150, 266666, 1, True, {"rows_lo": 1, "rows_hi": 10}),
    # bench.py's `dnasegment150`: dnasegment's DEFAULT block size (150 stations per block, include/config/dnaoptions.hpp:382): 667 blocks of
    # 150 inner + 50 junction stations, n = 600 -- whole on the CPU in under a minute
    "dnasegment150": (2000, 50, 266666, 1, True, {"rows_lo": 3, "rows_hi": 3}),
}


def synth_args(workload):
    """(rows, cols, baselines, blocks, phased, extra generator arguments) of a workload"""
    w = WORKLOADS[workload]
    return w[0], w[1], w[2], w[3], w[4], (w[5] if len(w) > 5 else {})
SEED = 20260928
ROW_STRIDE = 8          # rows kept of every sampled column


def sample_columns(n):
    """columns of a block's variance matrix that are kept: spread over the matrix, never 0 (a constrained corner)"""
    return sorted({min(n - 1, max(1, (n * p) // 97)) for p in (11, 43, 83)})


def sample_packed(ap, n):
    """ap: packed lower triangle, column-major (matrix_2d::packed_index).  Returns (diagonal, columns x rows[::ROW_STRIDE])"""
    ap = np.asarray(ap)
    j = np.arange(n, dtype=np.int64)
    col0 = j * n - j * (j - 1) // 2              # index of element (j, j)
    diag = ap[col0].copy()
    rows = np.arange(0, n, ROW_STRIDE, dtype=np.int64)
    cols = []
    for c in sample_columns(n):
        lo = np.minimum(rows, c)
        hi = np.maximum(rows, c)
        cols.append(ap[col0[lo] + (hi - lo)])
    return diag, np.stack(cols)


QUADRATIC_FORMS = 4
PREC_STRIDE = 50        # every 50th GNSS vector's adjusted-measurement precision (6 values: xx xy xz yy yz zz)


def packed_checksums(ap, n, block):
    """Sums over EVERY element of a block's packed variance matrix (the sampled diagonal / columns pin a few thousand of its 2e8 elements):
    the Frobenius norm of the symmetric matrix and QUADRATIC_FORMS quadratic forms x^T S x with seeded x in [-1, 1) -- any element
    that is wrong by more than the tolerance times the matrix's scale moves at least one of them."""
    ap = np.asarray(ap)
    j = np.arange(n, dtype=np.int64)
    col0 = j * n - j * (j - 1) // 2
    diag = ap[col0]
    fro = float(np.sqrt(2.0 * np.dot(ap, ap) - np.dot(diag, diag)))
    rng = np.random.default_rng(SEED + 1000 * block)
    X = rng.random((QUADRATIC_FORMS, n)) * 2.0 - 1.0
    q = np.zeros(QUADRATIC_FORMS)
    # x^T S x = sum_j x_j ( 2 * S[j+1:, j] . x[j+1:] + S[j, j] x_j ), column by column of the packed triangle
    for c in range(n):
        col = ap[col0[c]:col0[c] + n - c]
        q += X[:, c] * (2.0 * (X[:, c + 1:] @ col[1:]) + col[0] * X[:, c])
    return fro, q


def sample_precisions(prec, n_vectors):
    """prec: v_precAdjMsrsFull_ of a block (6 values per GNSS vector in CML order) -> the rows of every PREC_STRIDE-th vector"""
    p = np.asarray(prec)[:6 * n_vectors].reshape(-1, 6)
    return p[::PREC_STRIDE].copy()
