#!/usr/bin/env python
"""Full-size golden record of the CPU oracle (oracle/dna_oracle.c with the MKL runtime: the restated reference path) on one of
the bench's workloads:

    python tools/make_fullsize_golden.py cfg3 [out.npz]

cfg3 (100 172 stations, 16 blocks of n ~ 20 000, 92 Solve() calls of n^3 flops) needs ~80 GB of host memory and 7.4e14 flops:
run it where those exist (the GPU box's host: `gpurun -- python tools/make_fullsize_golden.py cfg3 gpurun_out/cfg3_oracle.npz`),
then commit the result as tests/golden/<workload>_oracle.npz.  tests/test_gpu_fullsize.py compares the device path with it.
The record: every adjusted coordinate, the diagonal of every block's rigorous variance matrix, three of its columns (every
8th row), its Frobenius norm and four seeded quadratic forms (sums over EVERY element), the precisions of every 50th adjusted
GNSS measurement (configs[4]'s consumer of the variances, ComputePrecisionAdjMsrs ADJ:7784), iteration count, per-iteration largest corrections, chi-squared / sigma-zero / degrees of freedom, and how long the
oracle took on how many threads."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")

import numpy as np


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", f"{workload}_oracle.npz")
    threads = int(os.environ.get("ORACLE_THREADS", "0")) or min(os.cpu_count() or 1, 64)
    from dynadjust_amd import adjust
    from tests import fullsize, oracle
    rows, cols, nbl, blocks, phased, kw = fullsize.synth_args(workload)
    d = tempfile.mkdtemp(prefix="dnagpu_golden_")
    info = adjust.write_synthetic_network(d, "net", rows, cols, nbl, blocks, seed=fullsize.SEED, **kw)
    # the faster of the host's LAPACKs (dpotrf + dpotri at n = 4 096): the MKL runtime the reference links, or the OpenBLAS inside the scipy
    # wheel -- on a non-Intel host MKL can be several times slower
    lib = oracle.load()
    rng = np.random.default_rng(0)
    A = rng.standard_normal((4096, 64))
    M0 = np.asfortranarray(A @ A.T + np.eye(4096) * 4096)
    best = None
    for name, path in (("MKL runtime (libmkl_rt)", oracle.MKL), ("OpenBLAS (scipy wheel)", oracle.scipy_openblas_path())):
        if os.environ.get("ORACLE_LAPACK") and os.environ["ORACLE_LAPACK"].lower() not in name.lower():
            continue
        if not path or not oracle.use_lapack(path):
            continue
        lib.orc_set_threads(threads)
        for rep in range(2):
            M = M0.copy(order="F")
            t0 = time.perf_counter()
            lib.orc_potrf_lower(4096, M.ctypes.data_as(oracle.f64p), 4096)
            lib.orc_potri_lower(4096, M.ctypes.data_as(oracle.f64p), 4096)
            dt = time.perf_counter() - t0
        print(f"{name}: {4096 ** 3 / dt / 1e12:.3f} TFLOP/s at n = 4096, {threads} threads", file=sys.stderr, flush=True)
        if best is None or dt < best[2]:
            best = (name, path, dt)
    if best is None:
        raise SystemExit("a threaded LAPACK is needed at this size")
    lapack_name = best[0]
    oracle.use_lapack(best[1])
    lib.orc_set_threads(threads)
    net = oracle.Network(os.path.join(d, "net"), phased)
    o = oracle.Adjustment(net, phased, threads=0)        # (thread count already set on the chosen library)
    o.prepare()
    t0 = time.perf_counter()
    status = o.run()
    dt = time.perf_counter() - t0
    solves, n3 = o.solve_stats()
    rec = {"status": status, "iterations": o.iterations(), "stations": info["stations"], "blocks": o.n_blocks,
           "corrections": [o.max_correction(i + 1) for i in range(o.iterations())]}
    arrays = {}
    st, _ = o.statistics()           # (also fills the precisions of the adjusted measurements, ComputePrecisionAdjMsrs ADJ:7784)
    for b in range(o.n_blocks):
        stn = o.block_stations(b)
        est = o.block_estimates(b)
        n = est.size
        var = o.block_variances(b)
        diag, colsamp = fullsize.sample_packed(var, n)
        fro, quad = fullsize.packed_checksums(var, n, b)
        prec = o.block_prec_adj_msrs(b)
        arrays[f"stations_{b}"] = stn.astype(np.uint32)
        arrays[f"estimates_{b}"] = est
        arrays[f"vdiag_{b}"] = diag
        arrays[f"vcols_{b}"] = colsamp
        arrays[f"vfro_{b}"] = np.array([fro])
        arrays[f"vquad_{b}"] = quad
        arrays[f"prec_{b}"] = fullsize.sample_precisions(prec, prec.size // 6)
        del var
    rec.update(chi_squared=st.chi_squared, sigma_zero=st.sigma_zero, dof=st.dof, oracle_seconds=dt, oracle_threads=threads,
               oracle_solves=int(solves), oracle_sum_n3=n3, oracle_tflops=n3 / dt / 1e12, cpu_count=os.cpu_count(),
               lapack=lapack_name)
    arrays["meta"] = np.frombuffer(json.dumps(rec).encode(), dtype=np.uint8)
    np.savez(out, **arrays)
    print(json.dumps(rec), flush=True)
    o.close()


if __name__ == "__main__":
    main()
