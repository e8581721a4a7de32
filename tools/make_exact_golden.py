#!/usr/bin/env python
"""tests/golden/exact_3k.npz: the EXTENDED-PRECISION solution (tests/exact.py: numpy longdouble, 64-bit mantissa, its own blocked Cholesky --
no LAPACK, nothing shared with the oracle or the device path but the file reader) of a 3-block phased GNSS chain at a size where the device's
128-tile kernel, its batched path and junctions of J = 300 unknowns are at work: 12 x 100 stations, n = 3 600 unknowns, blocks of n = 1 500,
four corner stations held with weights of 1e12.  Minutes on the CPU (longdouble matrix products run at ~0.5 GFLOP/s):

    python tools/make_exact_golden.py [out.npz]

The record: every coordinate as a double-double (hi + lo), and of every block's variance matrix (the block's stations out of the inverse of
the whole network's normals) the diagonal, three sampled columns, the Frobenius norm and four seeded quadratic forms (tests/fullsize.py: sums
over every element).  tests/test_gpu_exact.py compares device and oracle with it at 1e-8 m / 1e-8 relative; tests/test_oracle_adjust.py the oracle."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")

import numpy as np

SPEC = {"rows": 12, "cols": 100, "blocks": 3, "seed": 5151}


def pack_lower(M):
    n = M.shape[0]
    return np.concatenate([M[j:, j] for j in range(n)])


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "exact_3k.npz")
    from dynadjust_amd import adjust
    from tests import dnaformats as F, exact, fullsize, oracle
    d = tempfile.mkdtemp(prefix="dnagpu_exact_")
    info = adjust.write_synthetic_network(d, "e", SPEC["rows"], SPEC["cols"], 0, SPEC["blocks"], seed=SPEC["seed"])
    net = oracle.Network(os.path.join(d, "e"), False)
    t0 = time.perf_counter()
    x, V, its = exact.solve_sparse(net)
    dt = time.perf_counter() - t0
    ISL, JSL, CML, nets = F.read_seg(os.path.join(d, "e.seg"))
    arrays = {}
    hi = np.asarray(x, dtype=np.float64)
    arrays["x_hi"] = hi
    arrays["x_lo"] = np.asarray(x - hi.astype(np.longdouble), dtype=np.float64)
    for b in range(len(ISL)):
        stn = np.sort(np.concatenate([ISL[b], JSL[b]])).astype(np.int64)
        idx = (3 * stn[:, None] + np.arange(3)).ravel()
        ap = pack_lower(np.asarray(V[np.ix_(idx, idx)], dtype=np.float64))
        n = idx.size
        diag, cols = fullsize.sample_packed(ap, n)
        fro, quad = fullsize.packed_checksums(ap, n, b)
        arrays[f"stations_{b}"] = stn.astype(np.uint32)
        arrays[f"vdiag_{b}"] = diag
        arrays[f"vcols_{b}"] = cols
        arrays[f"vfro_{b}"] = np.array([fro])
        arrays[f"vquad_{b}"] = quad
    rec = dict(SPEC, stations=info["stations"], baselines=info["baselines"], unknowns=3 * info["stations"], iterations=its,
               junction_unknowns=[int(3 * len(j)) for j in JSL], block_unknowns=[int(3 * (len(i) + len(j))) for i, j in zip(ISL, JSL)],
               seconds=dt, longdouble_bits=int(np.finfo(np.longdouble).nmant) + 1)
    arrays["meta"] = np.frombuffer(json.dumps(rec).encode(), dtype=np.uint8)
    np.savez(out, **arrays)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
