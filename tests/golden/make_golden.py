"""Generates the committed fixtures in tests/golden/ (run once in the authoring container).

1. matrix_golden.json  -- the known-answer data of the reference's own matrix tests
   (/root/reference/tests/test_matrix.cpp: lines 196-221, 497-533, 535-568, 741-775, 1006-1025,
   1233-1259, 1279-1309), restated as inputs + expected outputs.  Only data is kept.
2. lapack_golden.npz   -- packed SPD matrices (constraint-like diagonal spikes, cond up to ~1e14)
   with their inverse from LAPACK dpotrf + dpotri (scipy's LAPACK), the routines
   matrix_2d::cholesky_inverse calls (dnamatrix_contiguous.cpp:982-984), with and without the
   scale_normals_to_unity wrapper of dna_adjust::Solve (dnaadjust.cpp:6614-6645).
3. tiny_net.{bst,bms,asl,seg,truth} + tiny_net_expected.npz -- a 12-station, 3-block GNSS chain
   written by the product's synthetic generator, with the rigorous coordinates / variances the CPU
   oracle produced for it (phased and simultaneous).
"""
import json
import os
import sys

import numpy as np
from scipy.linalg import lapack

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def pack(M):
    n = M.shape[0]
    return np.concatenate([M[j:, j] for j in range(n)])


def lapack_inverse(M):
    c, info = lapack.dpotrf(M, lower=1)
    assert info == 0
    inv, info = lapack.dpotri(c, lower=1)
    assert info == 0
    inv = np.tril(inv)
    return inv + np.tril(inv, -1).T


def main():
    kat3 = [[4.0, -1.0, -1.0], [-1.0, 3.0, -1.0], [-1.0, -1.0, 2.0]]
    kat3_inv = [[0.384615, 0.230769, 0.307692], [0.230769, 0.538462, 0.384615], [0.307692, 0.384615, 0.846154]]
    spd4 = [[5, 1, 2, 0], [1, 4, 1, 1], [2, 1, 6, 1], [0, 1, 1, 3]]
    B4 = [[1.0, 5.0], [2.0, 6.0], [3.0, 7.0], [4.0, 8.0]]
    spd5 = [[10, 1, 2, 0, 1], [1, 8, 1, 2, 0], [2, 1, 7, 1, 1], [0, 2, 1, 6, 1], [1, 0, 1, 1, 5]]
    golden = {
        "source": "reference tests/test_matrix.cpp (data only)",
        "cholesky_inverse_3x3": {"matrix": kat3, "inverse": kat3_inv, "tol": 1e-4, "ref": "test_matrix.cpp:196-221, 497-533"},
        "indefinite_2x2": {"matrix": [[2.0, 1.0], [1.0, -1.0]], "ref": "test_matrix.cpp:535-551"},
        "singular_2x2": {"matrix": [[1.0, 2.0], [2.0, 4.0]], "ref": "test_matrix.cpp:553-568"},
        "multiply_sym_4x4": {"matrix": spd4, "rhs": B4, "product": (np.array(spd4, float) @ np.array(B4)).tolist(), "tol": 1e-12,
                             "ref": "test_matrix.cpp:741-775"},
        "packed_end_to_end_3x3": {"matrix": kat3, "rhs": [10.0, 20.0, 30.0], "tol": 1e-12, "ref": "test_matrix.cpp:1233-1259"},
        "packed_5x5": {"matrix": spd5, "tol": 1e-12, "ref": "test_matrix.cpp:1279-1309"},
    }
    with open(os.path.join(HERE, "matrix_golden.json"), "w") as f:
        json.dump(golden, f, indent=1)

    rng = np.random.default_rng(20260928)
    out = {}
    for n in (3, 6, 129, 300):
        A = rng.standard_normal((n, n + 5))
        M = A @ A.T / n + np.eye(n)
        spike = np.ones(n)
        spike[:: max(3, n // 9)] = 1e12   # sigma_fixed = 1e-6 m  ->  weight 1e12 (dnaadjust.cpp:232-245)
        M = M + np.diag(spike - 1.0)
        M = (M + M.T) / 2
        out[f"ap_{n}"] = pack(M)
        out[f"inv_{n}"] = pack(lapack_inverse(M))
        s = 1.0 / np.sqrt(np.diag(M))
        Ms = M * np.outer(s, s)
        out[f"inv_scaled_{n}"] = pack(lapack_inverse(Ms) * np.outer(s, s))
        x = rng.standard_normal(n)
        out[f"x_{n}"] = x
        out[f"Ax_{n}"] = M @ x
    np.savez_compressed(os.path.join(HERE, "lapack_golden.npz"), **out)

    # tiny network + oracle results
    from dynadjust_amd import adjust
    from tests import oracle
    info = adjust.write_synthetic_network(HERE, "tiny_net", 4, 3, 0, 3)
    assert info["stations"] == 12
    oracle.use_mkl(True)
    res = {}
    for phased in (False, True):
        net = oracle.Network(os.path.join(HERE, "tiny_net"), phased)
        a = oracle.Adjustment(net, phased)
        a.prepare()
        st = a.run()
        tag = "phased" if phased else "simult"
        res[f"{tag}_status"] = st
        res[f"{tag}_iterations"] = a.iterations()
        res[f"{tag}_corrections"] = np.array([a.max_correction(i + 1) for i in range(a.iterations())])
        for b in range(a.n_blocks):
            res[f"{tag}_stations_{b}"] = a.block_stations(b)
            res[f"{tag}_estimates_{b}"] = a.block_estimates(b)
            res[f"{tag}_variances_{b}"] = a.block_variances(b)
        a.close()
    np.savez_compressed(os.path.join(HERE, "tiny_net_expected.npz"), **res)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
