"""Parity at BASELINE.json's full sizes (configs[1] cfg2: one block of n = 30 000; configs[2] cfg3: 100 172 stations, 16 blocks
of n ~ 20 000), where nearly all flops go through the 128-tile throughput kernel (gemm_f64_dma_kernel):

  * cfg3 on the device against the committed record of the CPU oracle's run of the same network (tests/golden/cfg3_oracle.npz,
    made by tools/make_fullsize_golden.py: every coordinate, every variance diagonal, sampled variance columns);
  * cfg3, condensed schedule (a.schur_carry, kept factors) against the reference's forward / reverse / combine schedule on the
    device: every coordinate, every element of every block's variance matrix;
  * cfg2 against the oracle run live (one n = 30 000 dpotrf + dpotri with the MKL runtime on the host);
  * cfg4-sized blocks (n ~ 27 000, 1 000-station junction rows): staged / budget-limited paths at the real size.

Tolerances (BASELINE.json): coordinates 1e-8 m; variances 1e-8 of the block's largest element."""
import json
import os

import numpy as np
import pytest

from dynadjust_amd import adjust
from tests import dnaformats as F
from tests import fullsize

pytestmark = pytest.mark.gpu

TOL_X = 1e-8
TOL_V = 1e-8


def _write(tmp_path, workload):
    rows, cols, nbl, blocks, phased, kw = fullsize.synth_args(workload)
    info = adjust.write_synthetic_network(str(tmp_path), "net", rows, cols, nbl, blocks, seed=fullsize.SEED, **kw)
    return info, phased


def _run(tmp_path, phased, **kw):
    p = adjust.ProjectSettings("net", str(tmp_path), adjust_mode=adjust.PhasedMode if phased else adjust.SimultaneousMode, **kw)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    st = a.AdjustNetwork()
    return a, st


def _record(path, rec):
    """parity figures of this run, for profiles/ (gpurun_out/ is merged back from the GPU box)"""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(rec, open(os.path.join(out, path), "w"), indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("workload", ["cfg3q", "cfg4q", "smallblocks", "dnasegment150", "cfg3"])
def test_against_the_oracle_record(built, golden_dir, tmp_path, workload):
    """the device path (condensed schedule, kept factors, four chains) against the committed record of the CPU oracle's run of the
    same network: cfg3q = four of cfg3's sixteen strips at cfg3's block size (n ~ 20 000: the oracle's run fits a 64 GB host),
    cfg3 = the whole of it (when its record has been made: ~80 GB and 7.4e14 flops on the CPU); cfg4q = four of cfg4's 128 strips at cfg4's
    block geometry (n ~ 27 000 with junction rows of 1 000 stations: J = 3 000, condensed blocks of 6 000 unknowns; 30 Solve() calls);
    smallblocks = bench.py's dnasegment-like cut of 100 200 stations into 120 blocks of n = 900 ... 4 950, whole (bucketed batches);
    dnasegment150 = the reference's DEFAULT block size (150 stations per block, dnaoptions.hpp:382; dnasegment.cpp:594-596): 100 000 stations
    in 666 blocks of n = 600, whole, four iterations"""
    path = os.path.join(golden_dir, f"{workload}_oracle.npz")
    if not os.path.exists(path):
        assert workload != "cfg3q", "tests/golden/cfg3q_oracle.npz is missing: python tools/make_fullsize_golden.py cfg3q"
        pytest.skip(f"no record of the oracle for {workload} (tools/make_fullsize_golden.py {workload})")
    g = np.load(path)
    meta = json.loads(bytes(g["meta"]).decode())
    info, phased = _write(tmp_path, workload)
    assert info["stations"] == meta["stations"]
    a, st = _run(tmp_path, phased, multi_thread=True)
    assert st == meta["status"] and a.CurrentIteration() == meta["iterations"]
    dcorr = max(abs(a.GetIterationCorrection(i + 1) - c) for i, c in enumerate(meta["corrections"]))
    assert dcorr < TOL_X
    a.GenerateStatistics()          # (also: the precisions of the adjusted measurements from the resident variances, configs[4])
    dx = dv = dvc = dfro = dquad = dprec = 0.0
    checksums = "vfro_0" in g.files
    for b in range(a.blockCount()):
        assert np.array_equal(a.block_stations(b), g[f"stations_{b}"])
        est = a.block_estimates(b)
        dx = max(dx, float(np.abs(est - g[f"estimates_{b}"]).max()))
        var = a.block_variances_packed(b)
        diag, cols = fullsize.sample_packed(var, est.size)
        scale = float(np.abs(g[f"vdiag_{b}"]).max())
        dv = max(dv, float(np.abs(diag - g[f"vdiag_{b}"]).max()) / scale)
        dvc = max(dvc, float(np.abs(cols - g[f"vcols_{b}"]).max()) / scale)
        if checksums:
            # sums over EVERY element of the matrix, and the consumer of the variances: A S A^T of every 50th GNSS measurement
            fro, quad = fullsize.packed_checksums(var, est.size, b)
            dfro = max(dfro, abs(fro - float(g[f"vfro_{b}"][0])) / float(g[f"vfro_{b}"][0]))
            dquad = max(dquad, float(np.abs(quad - g[f"vquad_{b}"]).max() / np.abs(g[f"vquad_{b}"]).max()))
            p_all = a.block_prec_adj_msrs(b)
            prec = fullsize.sample_precisions(p_all, p_all.size // 6)
            assert prec.shape == g[f"prec_{b}"].shape
            dprec = max(dprec, float(np.abs(prec - g[f"prec_{b}"]).max() / np.abs(g[f"prec_{b}"]).max()))
        del var
    rec = {"workload": workload, "stations": info["stations"], "blocks": a.blockCount(), "iterations": a.CurrentIteration(),
           "schedule": "condensed + kept factors, four chains", "max_abs_dx_m": dx, "max_rel_dvar_diagonal": dv,
           "max_rel_dvar_sampled_columns": dvc, "max_abs_dcorrection_m": dcorr,
           "every_element": ({"max_rel_dfrobenius": dfro, "max_rel_dquadratic_forms": dquad, "max_rel_dprecision_adjusted_measurements": dprec} if checksums else None),
           "sigma_zero_device": a.GetSigmaZero(), "sigma_zero_oracle": meta["sigma_zero"],
           "chi_squared_device": a.GetChiSquared(), "chi_squared_oracle": meta["chi_squared"],
           "oracle": {k: meta[k] for k in ("oracle_seconds", "oracle_threads", "oracle_tflops", "lapack", "cpu_count") if k in meta}}
    rec["unknowns_per_block"] = [int(g[f"estimates_{b}"].size) for b in range(a.blockCount())]
    _record(f"parity_{workload}.json", rec)
    assert dx < TOL_X and dv < TOL_V and dvc < TOL_V, rec
    assert dfro < TOL_V and dquad < 1e-7 and dprec < 1e-7, rec          # (a quadratic form sums 2e8 terms of either sign: 1e-7 of the largest form)
    if workload == "cfg3q":
        assert checksums, "tests/golden/cfg3q_oracle.npz predates the every-element checksums: python tools/make_fullsize_golden.py cfg3q"
    assert a.GetDegreesOfFreedom() == meta["dof"]
    assert abs(a.GetChiSquared() - meta["chi_squared"]) / meta["chi_squared"] < 1e-7
    a.close()


def test_cfg3_condensed_schedule_equals_reference_schedule(built, tmp_path):
    """a.schur_carry = 1 (condensed blocks, kept factors, four chains) against a.schur_carry = 0 (every forward / reverse /
    combination step a full inverse, the reference's schedule), both on the device at full size: every element"""
    info, phased = _write(tmp_path, "cfg3")
    a, st = _run(tmp_path, phased, multi_thread=True, schur_carry=True)
    assert st == adjust.ADJUST_SUCCESS and a.elimination_count() > 0 and a.completion_count() > 0
    B = a.blockCount()
    est = [a.block_estimates(b) for b in range(B)]
    var = [a.block_variances_packed(b) for b in range(B)]
    its = a.CurrentIteration()
    a.close()
    r, st = _run(tmp_path, phased, multi_thread=True, schur_carry=False)
    assert st == adjust.ADJUST_SUCCESS and r.elimination_count() == 0 and r.CurrentIteration() == its
    dx = dv = 0.0
    for b in range(B):
        dx = max(dx, float(np.abs(r.block_estimates(b) - est[b]).max()))
        v = r.block_variances_packed(b)
        dv = max(dv, float(np.abs(v - var[b]).max() / np.abs(v).max()))
        var[b] = None
    _record("parity_cfg3_schedules.json", {"workload": "cfg3", "max_abs_dx_m": dx, "max_rel_dvar": dv, "iterations": its,
                                           "compared": "condensed + kept factors vs reference schedule, every element"})
    assert dx < TOL_X and dv < TOL_V, (dx, dv)
    r.close()


def test_cfg2_against_the_oracle(built, orc, tmp_path):
    """BASELINE.json configs[1]: 10 000 stations, simultaneous, one n = 30 000 inverse -- against the oracle with a threaded LAPACK behind its
    dpotrf / dpotri (the reference links whichever the host has: MKL, OpenBLAS ... -- here the OpenBLAS of the scipy wheel where present: on
    the pool's non-Intel hosts the MKL runtime is several times slower, 97 s of the suite), every coordinate and every variance element"""
    info, phased = _write(tmp_path, "cfg2")
    fast = orc.scipy_openblas_path()
    if not ((fast and orc.use_lapack(fast)) or orc.use_mkl(True)):
        pytest.skip("needs a threaded LAPACK: the built-in Cholesky takes hours at n = 30 000")
    try:
        orc.load().orc_set_threads(min(os.cpu_count() or 1, 64))
        net = orc.Network(str(tmp_path / "net"), phased)
        o = orc.Adjustment(net, phased)
        o.prepare()
        ost = o.run()
    finally:
        orc.use_mkl(False)
    a, st = _run(tmp_path, phased)
    assert st == ost and a.CurrentIteration() == o.iterations()
    for i in range(o.iterations()):
        assert abs(a.GetIterationCorrection(i + 1) - o.max_correction(i + 1)) < TOL_X
    assert np.array_equal(a.block_stations(0), o.block_stations(0))
    dx = float(np.abs(a.block_estimates(0) - o.block_estimates(0)).max())
    vo = o.block_variances(0)
    dv = float(np.abs(a.block_variances_packed(0) - vo).max() / np.abs(vo).max())
    _record("parity_cfg2.json", {"workload": "cfg2", "unknowns": int(a.GetUnknownsCount()), "max_abs_dx_m": dx, "max_rel_dvar": dv,
                                 "iterations": a.CurrentIteration()})
    assert dx < TOL_X and dv < TOL_V, (dx, dv)
    a.close()
    o.close()


def test_cfg4_sized_blocks(built, tmp_path):
    """4 of cfg4's 128 strips (n ~ 27 000 per block, 1 000-station junction rows) on one GPU: the condensed schedule with the
    kept-factor budget and the staging decision at the real block size.  Size-independent properties: phased (condensed)
    == reference schedule on the device to 1e-8 m; staged == resident bit for bit; sigma-zero inside its 95 % limits."""
    rows, cols, nbl, blocks = 32, 1000, 85000, 4
    adjust.write_synthetic_network(str(tmp_path), "net", rows, cols, nbl, blocks, seed=fullsize.SEED)
    a, st = _run(tmp_path, True, multi_thread=True)
    assert st == adjust.ADJUST_SUCCESS
    B = a.blockCount()
    est = [a.block_estimates(b) for b in range(B)]
    vdiag = [fullsize.sample_packed(a.block_variances_packed(b), est[b].size) for b in range(B)]
    a.GenerateStatistics()
    assert a.GetChiSquaredLowerLimit() < a.GetSigmaZero() < a.GetChiSquaredUpperLimit()
    a.close()
    s, st = _run(tmp_path, True, multi_thread=True, stage=True)
    assert st == adjust.ADJUST_SUCCESS
    for b in range(B):
        assert np.array_equal(s.block_estimates(b), est[b])
        d, c = fullsize.sample_packed(s.block_variances_packed(b), est[b].size)
        assert np.array_equal(d, vdiag[b][0]) and np.array_equal(c, vdiag[b][1])
    s.close()
    r, st = _run(tmp_path, True, multi_thread=True, schur_carry=False)
    assert st == adjust.ADJUST_SUCCESS
    dx = dv = 0.0
    for b in range(B):
        dx = max(dx, float(np.abs(r.block_estimates(b) - est[b]).max()))
        d, c = fullsize.sample_packed(r.block_variances_packed(b), est[b].size)
        scale = float(np.abs(d).max())
        dv = max(dv, float(np.abs(d - vdiag[b][0]).max()) / scale, float(np.abs(c - vdiag[b][1]).max()) / scale)
    _record("parity_cfg4_blocks.json", {"blocks": B, "unknowns_per_block": int(est[1].size), "max_abs_dx_m": dx, "max_rel_dvar": dv})
    assert dx < TOL_X and dv < TOL_V, (dx, dv)
    r.close()


def test_cfg4_full_size_properties(built, tmp_path):
    """BASELINE.json configs[3] and [4] at FULL size on one GPU: 1 000 000 stations, 2 666 666 baselines (7 999 998 measurement rows), 128
    blocks of n ~ 27 000 -- one AdjustNetwork() to convergence in staged mode (373 GB of packed variance matrices: page-locked host memory up
    to the container's memory limit, the rest packed in HBM; blocks the HBM budget denies a kept factor make it again), then GenerateStatistics
    = configs[4]'s propagation of the variances to every adjusted measurement.  No oracle can run this in the test's time (764 Solve() calls
    of n^3 = 2e13: ~3 h of host LAPACK); cfg4's block geometry is pinned against the oracle by `cfg4q` above, and here the size-independent
    properties are checked: convergence, sigma-zero inside its 95 % limits at 5 000 010 degrees of freedom, every station within 0.25 m of the
    truth the measurements were drawn from, neighbouring blocks agreeing on their shared stations to 1e-8 m (the rigorous property of the
    phased adjustment, ADJ:2794-2796), variance matrices symmetric positive on their diagonals and equal on shared stations' diagonal blocks
    to 1e-8 relative, and the memory plan adding up."""
    rows, cols, nbl, blocks = 1000, 1000, 2666666, 128
    info = adjust.write_synthetic_network(str(tmp_path), "net", rows, cols, nbl, blocks)
    assert info["stations"] == 1000000 and info["measurement_rows"] == 7999998 and info["blocks"] == 128
    p = adjust.ProjectSettings("net", str(tmp_path), adjust_mode=adjust.PhasedMode, multi_thread=True, stage=True)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    plan = a.memory_plan()
    total_packed = sum((3 * a.lib.dnaadj_block_station_count(a.h, k)) * (3 * a.lib.dnaadj_block_station_count(a.h, k) + 1) // 2 * 8 for k in range(128))
    assert plan["staged_variances_host_bytes"] + plan["staged_variances_packed_in_hbm_bytes"] == total_packed
    if plan["staged_variances_host_bytes"] > 0.85 * plan["host_memory_available_gb"] * 1e9:
        a.close()
        pytest.skip("the host's memory limit leaves no margin for the staged store")
    import time
    t0 = time.perf_counter()
    st = a.AdjustNetwork()
    dt = time.perf_counter() - t0
    assert st == adjust.ADJUST_SUCCESS and a.CurrentIteration() <= 4
    after = a.memory_plan()
    a.GenerateStatistics()
    assert a.GetDegreesOfFreedom() == 7999998 - (3 * 1000000 - 12)
    assert a.GetChiSquaredLowerLimit() < a.GetSigmaZero() < a.GetChiSquaredUpperLimit()
    truth = np.fromfile(str(tmp_path / "net.truth"), dtype=np.float64).reshape(-1, 3)
    err = float(np.abs(a.adjusted_coordinates(1000000) - truth).max())
    assert err < 0.25
    # neighbouring blocks on their shared stations (estimates, and the 3 x 3 diagonal blocks of the variance matrices)
    dx = dv = 0.0
    for k in (0, 63, 126):
        s0, s1 = a.block_stations(k), a.block_stations(k + 1)
        x0, x1 = a.block_estimates(k).reshape(-1, 3), a.block_estimates(k + 1).reshape(-1, 3)
        common, i0, i1 = np.intersect1d(s0, s1, return_indices=True)
        assert 900 < common.size <= 1000            # (block k's junction stations: the first row of strip k + 1, where a measurement of block k ends)
        dx = max(dx, float(np.abs(x0[i0] - x1[i1]).max()))
        d0 = fullsize.sample_packed(a.block_variances_packed(k), 3 * s0.size)[0].reshape(-1, 3)
        d1 = fullsize.sample_packed(a.block_variances_packed(k + 1), 3 * s1.size)[0].reshape(-1, 3)
        assert d0.min() > 0 and d1.min() > 0
        dv = max(dv, float(np.abs(d0[i0] - d1[i1]).max() / d0.max()))
    _record("cfg4_full_size.json", {"stations": 1000000, "blocks": 128, "iterations": a.CurrentIteration(), "adjust_seconds": dt,
                                    "sigma_zero": a.GetSigmaZero(), "chi_squared_limits": [a.GetChiSquaredLowerLimit(), a.GetChiSquaredUpperLimit()],
                                    "degrees_of_freedom": a.GetDegreesOfFreedom(), "max_abs_error_vs_truth_m": err,
                                    "shared_stations_max_abs_dx_m": dx, "shared_stations_max_rel_dvar": dv, "memory_plan": after})
    assert dx < TOL_X and dv < TOL_V, (dx, dv)
    a.close()


def test_default_cut_project_of_a_million_stations(built, tmp_path):
    """a project of the national size at the reference's DEFAULT cut (dnasegment: 150 stations per block, dnaoptions.hpp:382): ten contiguous
    networks of 666 blocks (n = 600) each in one set of files -- 1 000 000 stations, 6 660 blocks, ten network ids (the networks generated
    independently and merged, tests/dnaformats.py::merge_networks).  The junction chains of all ten networks advance together in lock step
    (60 runs, dna_adjust::LockstepChains; one network's geometry is pinned against the oracle by `dnasegment150` above).  No oracle runs this in
    the test's time; the size-independent properties: convergence, sigma-zero inside its 95 % limits at 5 000 100 degrees of freedom, every
    station within 0.25 m of the truth, neighbouring blocks equal on their shared stations to 1e-8 m / 1e-8 relative, networks independent of
    each other (the first network's estimates are those of the same network adjusted alone, to rounding)."""
    parts = []
    for q in range(10):
        info = adjust.write_synthetic_network(str(tmp_path), f"part{q}", 2000, 50, 266666, 1, seed=20260930 + q, rows_lo=3, rows_hi=3)
        parts.append(str(tmp_path / f"part{q}"))
    F.merge_networks(parts, str(tmp_path / "net"))
    truth = np.concatenate([np.fromfile(p + ".truth", dtype=np.float64) for p in parts]).reshape(-1, 3)
    nb1, ns1 = info["blocks"], info["stations"]
    p = adjust.ProjectSettings("net", str(tmp_path), adjust_mode=adjust.PhasedMode, multi_thread=True)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    assert a.blockCount() == 10 * nb1 and a.chain_runs() >= 40
    import time
    t0 = time.perf_counter()
    st = a.AdjustNetwork()
    dt = time.perf_counter() - t0
    assert st == adjust.ADJUST_SUCCESS and a.CurrentIteration() <= 5
    a.GenerateStatistics()
    assert a.GetDegreesOfFreedom() == 10 * (3 * 266666 - (3 * ns1 - 12))
    assert a.GetChiSquaredLowerLimit() < a.GetSigmaZero() < a.GetChiSquaredUpperLimit()
    err = float(np.abs(a.adjusted_coordinates(10 * ns1) - truth).max())
    assert err < 0.25
    dx = dv = 0.0
    for k in (0, nb1 // 2, nb1 - 2, nb1, 5 * nb1 + 7, 10 * nb1 - 2):
        s0, s1 = a.block_stations(k), a.block_stations(k + 1)
        x0, x1 = a.block_estimates(k).reshape(-1, 3), a.block_estimates(k + 1).reshape(-1, 3)
        common, i0, i1 = np.intersect1d(s0, s1, return_indices=True)
        assert 40 <= common.size <= 50            # (the junction row of the next strip: the stations a measurement of this block ends at)
        dx = max(dx, float(np.abs(x0[i0] - x1[i1]).max()))
        d0 = fullsize.sample_packed(a.block_variances_packed(k), 3 * s0.size)[0].reshape(-1, 3)
        d1 = fullsize.sample_packed(a.block_variances_packed(k + 1), 3 * s1.size)[0].reshape(-1, 3)
        assert d0.min() > 0 and d1.min() > 0
        dv = max(dv, float(np.abs(d0[i0] - d1[i1]).max() / d0.max()))
    # the last block of one network and the first of the next share nothing
    assert np.intersect1d(a.block_stations(nb1 - 1), a.block_stations(nb1)).size == 0
    x_first = [a.block_estimates(k).copy() for k in (0, nb1 // 3, nb1 - 1)]
    iters = a.CurrentIteration()
    sigma = a.GetSigmaZero()
    a.close()
    # the first network alone
    b = adjust.DnaAdjust()
    b.PrepareAdjustment(adjust.ProjectSettings("part0", str(tmp_path), adjust_mode=adjust.PhasedMode, multi_thread=True))
    assert b.AdjustNetwork() == adjust.ADJUST_SUCCESS
    alone = max(float(np.abs(b.block_estimates(k) - x).max()) for k, x in zip((0, nb1 // 3, nb1 - 1), x_first))
    b.close()
    _record("default_cut_project_full_size.json", {"stations": 10 * ns1, "blocks": 10 * nb1, "networks": 10, "iterations": iters, "adjust_seconds": dt,
                                                   "sigma_zero": sigma, "max_abs_error_vs_truth_m": err, "shared_stations_max_abs_dx_m": dx,
                                                   "shared_stations_max_rel_dvar": dv, "first_network_alone_max_abs_dx_m": alone})
    assert dx < TOL_X and dv < TOL_V and alone < 5e-9, (dx, dv, alone)
