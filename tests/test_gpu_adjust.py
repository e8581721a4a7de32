"""End-to-end parity of the device path (through the dna_adjust facade and the C-ABI) with the CPU
oracle on identical input files.  Tolerances: estimated coordinates within 1e-8 m (BASELINE.json),
variance matrices within 1e-8 relative to their largest element."""
import os

import numpy as np
import pytest

from dynadjust_amd import adjust
from dynadjust_amd.device import unpack_lower
from tests import dnaformats as F

pytestmark = pytest.mark.gpu

TOL_X = 1e-8
TOL_V = 1e-8


def _device_run(folder, name, phased, **kw):
    p = adjust.ProjectSettings(name, folder, adjust_mode=adjust.PhasedMode if phased else adjust.SimultaneousMode, **kw)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    st = a.AdjustNetwork()
    return a, st


def _compare(a, st, o, ost):
    assert st == ost
    assert a.CurrentIteration() == o.iterations()
    for i in range(o.iterations()):
        assert abs(a.GetIterationCorrection(i + 1) - o.max_correction(i + 1)) < 1e-8
    for b in range(a.blockCount()):
        assert np.array_equal(a.block_stations(b), o.block_stations(b))
        assert np.abs(a.block_estimates(b) - o.block_estimates(b)).max() < TOL_X
        vo = o.block_variances(b)
        assert np.abs(a.block_variances_packed(b) - vo).max() / np.abs(vo).max() < TOL_V


def test_golden_tiny_network(built, golden_dir):
    exp = np.load(os.path.join(golden_dir, "tiny_net_expected.npz"))
    for phased, tag in ((False, "simult"), (True, "phased")):
        a, st = _device_run(golden_dir, "tiny_net", phased)
        assert st == int(exp[f"{tag}_status"]) and a.CurrentIteration() == int(exp[f"{tag}_iterations"])
        assert a.GetMeasurementCount() == 3 * 23 or a.GetMeasurementCount() > 0
        for b in range(a.blockCount()):
            assert np.array_equal(a.block_stations(b), exp[f"{tag}_stations_{b}"])
            assert np.abs(a.block_estimates(b) - exp[f"{tag}_estimates_{b}"]).max() < TOL_X
            v = exp[f"{tag}_variances_{b}"]
            assert np.abs(a.block_variances_packed(b) - v).max() / np.abs(v).max() < TOL_V
        a.close()


@pytest.mark.parametrize("rows,cols,nbl,blocks,phased,scale", [
    (6, 5, 0, 1, False, False),
    (12, 10, 300, 1, False, True),
    (12, 10, 300, 2, True, False),
    (12, 10, 300, 4, True, False),
    (12, 10, 300, 3, True, True),
    (9, 9, 160, 9, True, False),        # one grid row per block: every inner station is also a junction target
    (30, 30, 2400, 5, True, False),
    (45, 45, 0, 3, True, False),        # blocks larger than one 128x128 tile row in the junction matrices
])
def test_parity_with_oracle(built, orc, tmp_path, rows, cols, nbl, blocks, phased, scale):
    adjust.write_synthetic_network(str(tmp_path), "n", rows, cols, nbl, blocks, seed=rows + 31 * blocks)
    orc.use_mkl(True)
    try:
        net = orc.Network(str(tmp_path / "n"), phased)
        o = orc.Adjustment(net, phased, scale_normals_to_unity=scale)
        o.prepare()
        ost = o.run()
    finally:
        orc.use_mkl(False)
    a, st = _device_run(str(tmp_path), "n", phased, scale_normals_to_unity=scale)
    _compare(a, st, o, ost)
    assert a.GetUnknownsCount() == 3 * rows * cols - 12          # four CCC corner stations (dnaadjust.cpp:10567-10574)
    assert a.GetDegreesOfFreedom() == a.GetMeasurementCount() - a.GetUnknownsCount()
    a.close()
    o.close()


@pytest.mark.parametrize("rows,cols,nbl,blocks,phased,xcl,ycl", [
    (6, 6, 0, 1, False, 8, False),
    (7, 6, 0, 1, False, 1000, True),     # every station's baselines one 'X' cluster + 'Y' datum
    (12, 10, 0, 3, True, 40, True),
    (12, 10, 250, 4, True, 1000, False),
    (30, 30, 0, 5, True, 1000, True),
])
def test_gnss_cluster_parity_with_oracle(built, orc, tmp_path, rows, cols, nbl, blocks, phased, xcl, ycl):
    """'X' baseline clusters / 'Y' point clusters with full variance matrices (LoadVarianceMatrix_X/_Y, dnaadjust.cpp:4312/4494)"""
    adjust.write_synthetic_network(str(tmp_path), "c", rows, cols, nbl, blocks, seed=5 * rows + blocks, x_clusters=xcl, y_cluster=ycl)
    orc.use_mkl(True)
    try:
        net = orc.Network(str(tmp_path / "c"), phased)
        assert net.n_clusters > 0 and int(np.diff(net.cluster_off).max()) >= 2
        o = orc.Adjustment(net, phased)
        o.prepare()
        ost = o.run()
    finally:
        orc.use_mkl(False)
    a, st = _device_run(str(tmp_path), "c", phased)
    _compare(a, st, o, ost)
    assert a.GetMeasurementCount() == 3 * net.n_baselines
    assert a.GetUnknownsCount() == 3 * rows * cols - (0 if ycl else 12)
    a.close()
    o.close()


@pytest.mark.parametrize("rows,cols,blocks,xcl,ycl", [(10, 8, 3, 0, False), (9, 12, 4, 1000, True)])
def test_device_against_the_extended_precision_solution(built, orc, tmp_path, rows, cols, blocks, xcl, ycl):
    """the device path against the EXACT least-squares solution (tests/exact.py: numpy longdouble, dense, its own Cholesky): simultaneous and
    phased results within three units in the last place of a 4e6 m coordinate (2.8e-9 m), variances within 1e-11 relative -- the 1e-8 m
    claim argued through the exact answer instead of through another fp64 solver"""
    from tests import exact
    adjust.write_synthetic_network(str(tmp_path), "e", rows, cols, 0, blocks, seed=31 + rows, x_clusters=xcl, y_cluster=ycl)
    net = orc.Network(str(tmp_path / "e"), False)
    x, V, its = exact.solve(net)
    a, st = _device_run(str(tmp_path), "e", False)
    assert st == 0 and a.CurrentIteration() == its
    assert float(np.abs(np.asarray(a.block_estimates(0), dtype=np.longdouble) - x).max()) < 2.8e-9
    Vd = unpack_lower(a.block_variances_packed(0), 3 * net.n_stations)
    assert float(np.abs(np.asarray(Vd, dtype=np.longdouble) - V).max() / np.abs(V).max()) < 1e-11
    a.close()
    for mt in (False, True):
        p, st = _device_run(str(tmp_path), "e", True, multi_thread=mt)
        assert st == 0
        for b in range(p.blockCount()):
            stn = p.block_stations(b)
            xb = np.asarray(p.block_estimates(b), dtype=np.longdouble).reshape(-1, 3)
            assert float(np.abs(xb - x.reshape(-1, 3)[stn]).max()) < 2.8e-9
            idx = (3 * stn[:, None] + np.arange(3)).ravel()
            Vb = np.asarray(unpack_lower(p.block_variances_packed(b), 3 * len(stn)), dtype=np.longdouble)
            assert float(np.abs(Vb - V[np.ix_(idx, idx)]).max() / np.abs(V).max()) < 1e-11
        p.close()


def test_multiple_networks_and_isolated_blocks(built, orc, tmp_path):
    specs = [("a", 8, 5, 3), ("b", 5, 5, 1), ("c", 6, 6, 2)]
    for nm, r, c, blk in specs:
        adjust.write_synthetic_network(str(tmp_path), nm, r, c, 0, blk, seed=ord(nm))
    F.merge_networks([str(tmp_path / s[0]) for s in specs], str(tmp_path / "all"))
    net = orc.Network(str(tmp_path / "all"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "all", True)
    assert a.blockCount() == 6
    _compare(a, st, o, ost)
    a.close()
    o.close()


@pytest.mark.parametrize("seed", range(20))
def test_random_segmentations_against_the_oracle(built, orc, tmp_path, seed):
    """20 seeded block graphs that the strip generator cannot make (tests/segfuzz.py, tests/test_oracle_adjust.py::_fuzzed_project): junction
    stations that stay junction over 2 ... 5 and more blocks (dnasegment.cpp:529-531), uneven junction sets, blocks without measurements of
    their own, two network ids plus an isolated block (seg_file.cpp:305-392, dnaadjust.cpp:10449-10474) -- oracle live, device on both
    schedules (condensed, and the reference's own with a.schur_carry = 0), one and four chains"""
    from tests.test_oracle_adjust import _fuzzed_project
    info = _fuzzed_project(tmp_path, seed)
    net = orc.Network(str(tmp_path / "all"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    assert ost == 0
    for schur in (True, False):
        a, st = _device_run(str(tmp_path), "all", True, schur_carry=schur, multi_thread=bool((seed + schur) % 2))
        _compare(a, st, o, ost)
        if schur and a.CurrentIteration() >= 2 and a.condensed_schedule():
            assert a.factor_reuses() > 0              # (GNSS only: iterations >= 2 on the kept factors, persistent junctions included)
        a.close()
    o.close()


def test_iteration_limit_and_threshold(built, orc, golden_dir, tmp_path):
    a, st = _device_run(golden_dir, "tiny_net", True, max_iterations=1)
    assert st == adjust.ADJUST_MAX_ITERATIONS_EXCEEDED and a.CurrentIteration() == 1
    a.close()
    # an adjustment that runs out of iterations still ends with the variance matrices of its last iteration (they are formed after the
    # loop, a.defer_variances): same status, estimates and variances as the oracle stopped at the same point
    adjust.write_synthetic_network(str(tmp_path), "m", 14, 12, 0, 3, seed=5, initial_sigma=2.0)
    for limit in (1, 2):
        net = orc.Network(str(tmp_path / "m"), True)
        o = orc.Adjustment(net, True, max_iterations=limit)
        o.prepare()
        ost = o.run()
        a, st = _device_run(str(tmp_path), "m", True, max_iterations=limit)
        assert st == ost and (limit > 1 or st == adjust.ADJUST_MAX_ITERATIONS_EXCEEDED)
        _compare(a, st, o, ost)
        a.close()
        o.close()
    a, st = _device_run(golden_dir, "tiny_net", True, iteration_threshold=10.0)
    assert st == adjust.ADJUST_SUCCESS and a.CurrentIteration() == 1
    a.close()


def test_errors_follow_the_reference(built, tmp_path, golden_dir):
    # missing files
    a = adjust.DnaAdjust()
    with pytest.raises(adjust.NetAdjustException) as e:
        a.PrepareAdjustment(adjust.ProjectSettings("nothing", str(tmp_path)))
    assert "PrepareAdjustment(): Process terminated while preparing the" in str(e.value)
    a.close()
    # a singular block: an unconstrained... every station carries at least the free-station weight, so make the
    # normals indefinite through a negative-definite measurement variance instead
    import shutil
    for ext in ("bst", "bms", "asl", "seg"):
        shutil.copy(os.path.join(golden_dir, "tiny_net." + ext), str(tmp_path / ("bad." + ext)))
    bms = F.read_bms(str(tmp_path / "bad.bms")).copy()
    bms["term2"][0] = -1.0
    F.write_bms(str(tmp_path / "bad.bms"), bms)
    a = adjust.DnaAdjust()
    with pytest.raises(adjust.NetAdjustException) as e:
        a.PrepareAdjustment(adjust.ProjectSettings("bad", str(tmp_path), adjust_mode=adjust.PhasedMode))
    assert "singular" in str(e.value)
    a.close()
    # AdjustNetwork before PrepareAdjustment
    a = adjust.DnaAdjust()
    with pytest.raises(adjust.NetAdjustException):
        a.AdjustNetwork()
    a.close()


def test_phased_is_rigorous_at_scale(built, tmp_path):
    """size-independent property at a size the CPU oracle would need minutes for: the phased result on the
    device equals the simultaneous result on the device (12 800 unknowns, 8 blocks)"""
    adjust.write_synthetic_network(str(tmp_path), "big", 80, 80, 17000, 8, seed=5)
    s, st_s = _device_run(str(tmp_path), "big", False)
    p, st_p = _device_run(str(tmp_path), "big", True)
    assert st_s == 0 and st_p == 0
    xs = s.block_estimates(0).reshape(-1, 3)
    for b in range(p.blockCount()):
        stn = p.block_stations(b)
        assert np.abs(p.block_estimates(b).reshape(-1, 3) - xs[stn]).max() < TOL_X
    # variance of block 3 against the matching sub-matrix of the simultaneous inverse
    n = 3 * 80 * 80
    Vs = s.block_variances_packed(0)
    stn = p.block_stations(3)
    Vb = unpack_lower(p.block_variances_packed(3), 3 * len(stn))
    idx = (3 * stn[:, None] + np.arange(3)).ravel()
    # pick packed elements of the simultaneous matrix without unpacking 19200^2 doubles
    ii, jj = np.meshgrid(idx, idx, indexing="ij")
    lo, hi = np.minimum(ii, jj).astype(np.int64), np.maximum(ii, jj).astype(np.int64)
    sub = Vs[lo * n - lo * (lo - 1) // 2 + (hi - lo)]
    assert np.abs(Vb - sub).max() / np.abs(sub).max() < TOL_V
    s.close()
    p.close()


def test_multi_thread_mode_matches_oracle(built, orc, tmp_path):
    """--multi-thread: forward chain and reverse/combine chain on two streams of one GPU, two host threads
    (dnaadjust-multi.cpp:92-244); same results as the sequential schedule"""
    adjust.write_synthetic_network(str(tmp_path), "n", 36, 20, 0, 6, seed=77)
    net = orc.Network(str(tmp_path / "n"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "n", True, multi_thread=True)
    _compare(a, st, o, ost)
    # and bit-identical to the single-chain device run
    s, st_s = _device_run(str(tmp_path), "n", True)
    for b in range(a.blockCount()):
        assert np.array_equal(a.block_estimates(b), s.block_estimates(b))
    a.close()
    s.close()
    o.close()


@pytest.mark.parametrize("schur", [True, False])
def test_orchestrator_single_rank_equals_facade(built, tmp_path, schur):
    """tests/parallel_harness.run_phased on one rank drives the same per-block steps as AdjustPhased: the condensed
    schedule (a.schur_carry, default) and the reference's"""
    from tests import parallel_harness as parallel
    import torch
    adjust.write_synthetic_network(str(tmp_path), "n", 30, 12, 0, 5, seed=9)
    f, st_f = _device_run(str(tmp_path), "n", True, schur_carry=schur)
    p = adjust.ProjectSettings("n", str(tmp_path), adjust_mode=adjust.PhasedMode, schur_carry=schur)
    be = parallel.DeviceBlockBackend(p, torch.device("cpu"))
    assert be.condensed() == schur
    st, its, corr = parallel.run_phased(be, None, 0, 1)
    assert st == st_f and its == f.CurrentIteration()
    for b in range(f.blockCount()):
        assert np.array_equal(be.adj.block_estimates(b), f.block_estimates(b))
        assert np.array_equal(be.adj.block_variances_packed(b), f.block_variances_packed(b))
    be.close()
    f.close()


def _two_rank_worker(rank, world, port, folder, outdir, schur):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynadjust_amd import adjust as adj
    from tests import parallel_harness as parallel
    p = adj.ProjectSettings("n", folder, adjust_mode=adj.PhasedMode, schur_carry=schur)
    be = parallel.DeviceBlockBackend(p, torch.device("cpu"))     # both ranks share the box's single GPU; payloads via host
    st, its, corr = parallel.run_phased(be, dist, rank, world)
    if schur:
        owner = parallel.block_owners([float(be.n_stations(k)) ** 3 for k in range(be.n_blocks)], world)
        final_owner = lambda k: owner[k]
    else:
        final_owner = parallel.PhasedSchedule([be.flags(k) for k in range(be.n_blocks)], world).final_owner
    res = {"status": st, "iterations": its}
    # GenerateStatistics with the rigorous variances spread over the ranks
    parallel.distributed_statistics(be, dist, rank, world, final_owner)
    a = be.adj
    res["stats"] = np.array([a.GetChiSquared(), a.GetSigmaZero(), a.GetGlobalPelzerRel(), float(a.GetPotentialOutlierCount()),
                             float(a.GetDegreesOfFreedom()), float(a.GetTestResult())])
    res["records"] = np.frombuffer(a.measurement_records().tobytes(), dtype=np.uint8)
    for k in range(be.n_blocks):
        res[f"coords_{k}"] = be.get_coords(k)
        if final_owner(k) == rank:
            res[f"var_{k}"] = be.adj.block_variances_packed(k)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **res)
    dist.barrier()
    be.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("schur", [True, False])
def test_orchestrator_two_ranks_device_backend(built, orc, tmp_path, schur):
    """the real device backend under both 2-rank schedules (gloo transport, both processes on this box's GPU).  Condensed:
    blocks condensed and solved by their owners, condensed systems broadcast, chains everywhere.  Reference: junction
    export/import, combination solves on the 'other' rank.  Both: coordinate all_reduce"""
    import socket
    import torch.multiprocessing as mp
    adjust.write_synthetic_network(str(tmp_path), "n", 30, 12, 0, 6, seed=10)
    net = orc.Network(str(tmp_path / "n"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), str(tmp_path), schur), nprocs=2, join=True)
    # one process, the facade's own GenerateStatistics
    f, st_f = _device_run(str(tmp_path), "n", True, schur_carry=schur)
    f.GenerateStatistics()
    fstats = np.array([f.GetChiSquared(), f.GetSigmaZero(), f.GetGlobalPelzerRel(), float(f.GetPotentialOutlierCount()),
                       float(f.GetDegreesOfFreedom()), float(f.GetTestResult())])
    frec = np.frombuffer(f.measurement_records().tobytes(), dtype=F.MEASUREMENT_DT)
    f.close()
    seen = set()
    for r in range(2):
        res = np.load(str(tmp_path / f"rank{r}.npz"))
        # every rank ends with the whole network's statistics and every record's fields
        assert np.abs(res["stats"] - fstats).max() < 1e-9 * max(1.0, np.abs(fstats).max())
        rrec = np.frombuffer(res["records"].tobytes(), dtype=F.MEASUREMENT_DT)
        for nm in ("measAdj", "measCorr", "measAdjPrec", "residualPrec", "NStat", "PelzerRel"):
            assert np.abs(rrec[nm] - frec[nm]).max() <= 1e-9 * max(1.0, np.abs(frec[nm]).max()), nm
        assert int(res["status"]) == ost and int(res["iterations"]) == o.iterations()
        for k in range(6):
            assert np.abs(res[f"coords_{k}"] - o.block_estimates(k)).max() < TOL_X
            if f"var_{k}" in res:
                seen.add(k)
                vo = o.block_variances(k)
                assert np.abs(res[f"var_{k}"] - vo).max() / np.abs(vo).max() < TOL_V
    assert seen == set(range(6))
    o.close()


def _device_fields(a, bms_path):
    """per vector component arrays (network vector order) out of the records held by the adjustment"""
    raw = a.measurement_records()
    rec = np.frombuffer(raw.tobytes(), dtype=F.MEASUREMENT_DT)
    comp = rec[(rec["measStart"] < 3) & (~rec["ignore"])]
    out = {k: np.array(comp[k]) for k in ("measAdj", "measCorr", "measAdjPrec", "residualPrec", "NStat", "TStat", "PelzerRel")}
    out["measPrec"] = np.where(comp["measStart"] == 0, comp["term2"], np.where(comp["measStart"] == 1, comp["term3"], comp["term4"]))
    return out, rec


def _compare_statistics(a, o, confidence=95.0):
    from scipy.stats import chi2
    a.GenerateStatistics()
    so, fo = o.statistics(confidence)
    fd, rec = _device_fields(a, None)
    assert a.GetDegreesOfFreedom() == so.dof and a.GetMeasurementCount() == so.measurement_params
    # (1e-8 relative: the sum of squared residuals moves by 2.5e-9 of itself when the junction carry takes its information form --
    #  residuals of millimetres agreeing to 1e-12 m, three orders below an ulp of the coordinates they are differences of)
    assert abs(a.GetChiSquared() - so.chi_squared) < 1e-8 * max(1.0, so.chi_squared)
    assert abs(a.GetSigmaZero() - so.sigma_zero) < 1e-8 * max(1.0, so.sigma_zero)
    assert abs(a.GetGlobalPelzerRel() - so.global_pelzer) < 1e-7
    assert a.GetPotentialOutlierCount() == so.potential_outliers
    for k in ("measAdj", "measCorr"):
        assert np.abs(fd[k] - fo[k]).max() < 1e-8
    assert np.array_equal(fd["measPrec"], fo["measPrec"])
    scale = np.abs(fo["measAdjPrec"]).max()
    assert np.abs(fd["measAdjPrec"] - fo["measAdjPrec"]).max() < 1e-7 * scale
    assert np.abs(fd["residualPrec"] - fo["residualPrec"]).max() < 1e-7 * scale
    assert np.abs(fd["NStat"] - fo["NStat"]).max() < 1e-5
    ok = fo["PelzerRel"] < 100
    assert np.abs(fd["PelzerRel"][ok] - fo["PelzerRel"][ok]).max() < 1e-5
    half = (100.0 - confidence) * 0.005
    dof = so.dof
    assert abs(a.GetChiSquaredUpperLimit() - chi2.ppf(1 - half, dof) / dof) < 1e-9
    assert abs(a.GetChiSquaredLowerLimit() - chi2.ppf(half, dof) / dof) < 1e-9
    s0 = a.GetSigmaZero()
    assert a.GetTestResult() == (1 if s0 < a.GetChiSquaredLowerLimit() else (2 if s0 > a.GetChiSquaredUpperLimit() else 0))
    for b in range(a.blockCount()):
        po = o.block_prec_adj_msrs(b)
        assert np.abs(a.block_prec_adj_msrs(b) - po).max() < 1e-7 * np.abs(po).max()
    return fd, rec


def _against_oracle(orc, base, iterations, estimates, variances, corrections):
    """results of a device run (lists per block) against the oracle's adjustment of the same files"""
    net = orc.Network(base, True)
    o = orc.Adjustment(net, True)
    o.prepare()
    assert o.run() == 0 and o.iterations() == iterations
    for i, c in enumerate(corrections):
        assert abs(c - o.max_correction(i + 1)) < 1e-8
    for b in range(len(estimates)):
        assert np.abs(estimates[b] - o.block_estimates(b)).max() < TOL_X
        vo = o.block_variances(b)
        assert np.abs(variances[b] - vo).max() / np.abs(vo).max() < TOL_V
    o.close()


@pytest.mark.parametrize("importer", ["product", "test"])
def test_reference_sample_gnss_network(built, orc, golden_dir, tmp_path, importer):
    """the device path against the reference's published adjustment of its own sample network
    (sampleData/gnss-network.* -> gnss.simult.adj.expected) and against the oracle on the same files.
    importer "product": the .stn / .msr text goes through the product's importer (dnaimport_text: DNA reader + the frame alignment of
    dnareftran) -- the whole way from the reference's sample files to its report without anything taken from the report;
    "test": the test-only reader with the observations of the report's "Measured" column (round 1)"""
    from tests import dnatext as T
    from tests.test_oracle_adjust import check_against_reference_report
    base = str(tmp_path / "gnss")
    stn, cl, adj = (T.build_gnss_sample_with_the_product_importer if importer == "product" else T.build_gnss_sample)(golden_dir, base)
    net = orc.Network(base, False)
    o = orc.Adjustment(net, False)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "gnss", False, output_tstat=True)
    _compare(a, st, o, ost)
    assert st == 0 and a.CurrentIteration() == 2
    fd, rec = _compare_statistics(a, o)
    V = unpack_lower(a.block_variances_packed(0), 3 * len(stn))
    stats = {"measurements": a.GetMeasurementCount(), "unknowns": a.GetUnknownsCount(), "dof": a.GetDegreesOfFreedom(),
             "chi2": a.GetChiSquared(), "sigma0": a.GetSigmaZero(), "pelzer": a.GetGlobalPelzerRel(), "outliers": a.GetPotentialOutlierCount()}
    check_against_reference_report(adj, [x[0] for x in stn], stn, a.block_estimates(0), V, fd, stats)
    assert abs(a.GetChiSquaredUpperLimit() - 1.170) < 6e-4 and abs(a.GetChiSquaredLowerLimit() - 0.843) < 6e-4   # report: 0.843 < 1.169 < 1.170
    assert a.GetTestResult() == 0                                                                                 # "*** PASSED ***"
    assert np.abs(fd["TStat"] - fd["NStat"] / np.sqrt(a.GetSigmaZero())).max() < 1e-12
    a.close()
    o.close()


@pytest.mark.parametrize("rows,cols,blocks,phased,xcl,ycl,mt", [
    (8, 7, 1, False, 0, False, False),
    (9, 8, 3, True, 12, True, False),
    (12, 10, 4, True, 1000, False, True),
])
def test_statistics_parity_with_oracle(built, orc, tmp_path, rows, cols, blocks, phased, xcl, ycl, mt):
    adjust.write_synthetic_network(str(tmp_path), "s", rows, cols, 0, blocks, seed=rows, x_clusters=xcl, y_cluster=ycl)
    net = orc.Network(str(tmp_path / "s"), phased)
    o = orc.Adjustment(net, phased)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "s", phased, multi_thread=mt, confidence_interval=99.0)
    _compare(a, st, o, ost)
    _compare_statistics(a, o, 99.0)
    a.close()
    o.close()


_read_mtx_file = F.read_mtx


def test_results_out_files(built, orc, tmp_path):
    """SerialiseAdjustedVarianceMatrices (-rva.mtx / -pam.mtx) and UpdateBinaryFiles (.bst / .bms), dnaadjust.cpp:6770 / :445"""
    adjust.write_synthetic_network(str(tmp_path), "w", 9, 8, 0, 3, seed=3, x_clusters=6, y_cluster=True)
    bst0 = F.read_bst(str(tmp_path / "w.bst")).copy()
    a, st = _device_run(str(tmp_path), "w", True)
    assert st == 0
    a.GenerateStatistics()
    a.SerialiseAdjustedVarianceMatrices()
    rva = _read_mtx_file(str(tmp_path / "w-rva.mtx"), a.blockCount())
    pam = _read_mtx_file(str(tmp_path / "w-pam.mtx"), a.blockCount())
    for b in range(a.blockCount()):
        n = 3 * len(a.block_stations(b))
        assert rva[b][:3] == (1, n, n) and np.array_equal(rva[b][3], a.block_variances_packed(b))
        p = a.block_prec_adj_msrs(b)
        assert pam[b][:3] == (0, len(p), 1) and np.array_equal(pam[b][3], p) and len(p) > 0
    fd, rec = _device_fields(a, None)
    xyz = a.adjusted_coordinates(len(bst0))
    chi = a.GetChiSquared()
    a.UpdateBinaryFiles()
    a.close()
    # the files now hold the adjusted stations and measurements
    bst = F.read_bst(str(tmp_path / "w.bst"))
    bms = F.read_bms(str(tmp_path / "w.bms"))
    assert np.array_equal(np.frombuffer(bms.tobytes(), dtype=np.uint8), np.frombuffer(rec.tobytes(), dtype=np.uint8))
    for s in range(len(bst)):
        got = orc.geo_to_cart(float(bst["currentLatitude"][s]), float(bst["currentLongitude"][s]), float(bst["currentHeight"][s]))
        # CartToGeo (Lin & Wang, one Newton step, dnatemplategeodesyfuncs.hpp:154-225) is good to ~1e-7 m in height
        assert np.abs(np.array(got) - xyz[s]).max() < 1e-6
    assert np.array_equal(bst["initialLatitude"], bst0["initialLatitude"])
    # a second adjustment from the updated ("reduced") files starts at the solution: same statistics, tiny corrections
    a2, st2 = _device_run(str(tmp_path), "w", True)
    assert st2 == 0 and a2.CurrentIteration() == 1 and abs(a2.GetMaxCorrection()) < 1e-5
    a2.GenerateStatistics()
    assert abs(a2.GetChiSquared() - chi) < 1e-6 * chi
    a2.close()


@pytest.mark.parametrize("mt", [False, True])
def test_reuse_inverses_is_identical(built, orc, tmp_path, mt):
    """a.reuse_inverses (device path only): the block inverses of the first iteration are kept and reused -- for a GNSS-only
    network they are the same bits every iteration, so every result must be IDENTICAL, with half the Solve() calls"""
    adjust.write_synthetic_network(str(tmp_path), "r", 14, 12, 0, 5, seed=9, x_clusters=20, y_cluster=True, initial_sigma=0.4)
    runs = []
    for reuse in (False, True):
        a, st = _device_run(str(tmp_path), "r", True, multi_thread=mt, reuse_inverses=reuse, schur_carry=False)
        assert st == 0 and a.CurrentIteration() >= 2
        a.GenerateStatistics()
        runs.append((a.CurrentIteration(), a.solve_count(), [a.block_estimates(b) for b in range(a.blockCount())],
                     [a.block_variances_packed(b) for b in range(a.blockCount())], a.GetChiSquared(),
                     [a.GetIterationCorrection(i + 1) for i in range(a.CurrentIteration())]))
        a.close()
    (it0, n0, x0, v0, c0, corr0), (it1, n1, x1, v1, c1, corr1) = runs
    B = len(x0)
    assert it0 == it1 and corr0 == corr1 and c0 == c1
    assert n0 == it0 * (3 * B - 2) and n1 == 3 * B - 2
    _against_oracle(orc, str(tmp_path / "r"), it1, x1, v1, corr1)
    for b in range(B):
        assert np.array_equal(x0[b], x1[b]) and np.array_equal(v0[b], v1[b])
    # a second AdjustNetwork on the same object starts over (ResetAdjustment drops the resident inverses)
    a, st = _device_run(str(tmp_path), "r", True, multi_thread=mt, reuse_inverses=True, schur_carry=False)
    a.ResetAdjustment()
    assert a.AdjustNetwork() == 0 and a.solve_count() == 3 * B - 2
    for b in range(B):
        assert np.array_equal(a.block_estimates(b), x0[b])
    a.close()


@pytest.mark.parametrize("blocks,phased,ycl", [(1, False, True), (3, True, True), (4, True, False)])
def test_scalars_and_llh_point_clusters_parity(built, orc, tmp_path, blocks, phased, ycl):
    """variance scalars (LoadVarianceScaling, ScaleGPSVCV[_Cluster]) and Y clusters in latitude / longitude / height
    (PropagateVariances_GeoCart_Cluster, GeoToCart of the points): facade against oracle, including the statistics,
    which read the scaled variances back from the records (SetGPSVarianceMatrix)"""
    adjust.write_synthetic_network(str(tmp_path), "s", 10, 8, 0, blocks, seed=2, x_clusters=14, y_cluster=ycl, y_llh=ycl, scalars=True)
    net = orc.Network(str(tmp_path / "s"), phased)
    o = orc.Adjustment(net, phased)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "s", phased)
    _compare(a, st, o, ost)
    fd, rec = _compare_statistics(a, o)
    if ycl:
        y = rec[(rec["measType"] == b"Y") & (rec["measStart"] == 0)]
        assert {bytes(c).rstrip(b"\x00") for c in y["coordType"]} == {b"XYZ"}          # converted once and for all (ADJ:6313)
        assert set(int(v) for v in y["station3"]) == {1, 2}                              # LLh_type_i, LLH_type_i retained
        assert np.all(np.abs(y["preAdjMeas"]) < 4.0) and np.all(np.abs(y["term1"]) > 1e6)  # radians kept, metres now
    a.close()
    o.close()


@pytest.mark.parametrize("mt,blocks,terr", [(False, 5, False), (True, 5, False), (False, 2, False), (True, 3, True)])
def test_schur_carry_matches_full_solves(built, orc, tmp_path, mt, blocks, terr):
    """a.schur_carry (device path only, default on): the forward / reverse steps whose solution is only carried to the next
    block eliminate their inner unknowns instead of inverting the block.  Same estimates, variances and statistics as the
    reference's schedule of full solves (and as the oracle), with 2 (B - 1) of the 3 B - 2 solves per iteration replaced."""
    if terr:
        from tests import terrestrial_net as T
        T.build_mixed_network(str(tmp_path / "c"), 6, 4, blocks, seed=11)
    else:
        adjust.write_synthetic_network(str(tmp_path), "c", 16, 12, 0, blocks, seed=4, x_clusters=12, y_cluster=True, initial_sigma=0.3)
    runs = []
    # ... a.keep_factors: the rigorous solve completes the factor the condensing step kept; a.defer_variances (default): the iterations
    # take their corrections from the completed factor and the inverses are formed once, at the end (2: the light form of the factor)
    for schur, keep, defer in ((False, False, 0), (True, False, 0), (True, True, 0), (True, True, 1), (True, True, 2)):
        a, st = _device_run(str(tmp_path), "c", True, multi_thread=mt, schur_carry=schur, keep_factors=keep, defer_variances=defer)
        assert st == 0
        a.GenerateStatistics()
        # a.keep_factors: every block's rigorous solve completes the factor its condensing step kept
        assert a.completion_count() == (a.CurrentIteration() * a.blockCount() if keep else 0)
        runs.append((a.CurrentIteration(), a.solve_count(), a.elimination_count(), [a.block_estimates(b) for b in range(a.blockCount())],
                     [a.block_variances_packed(b) for b in range(a.blockCount())], a.GetChiSquared(), a.GetMaxCorrection()))
        if schur:
            net = orc.Network(str(tmp_path / "c"), True)
            o = orc.Adjustment(net, True)
            o.prepare()
            _compare(a, st, o, o.run())
            o.close()
        a.close()
    (it0, n0, e0, x0, v0, c0, mc0) = runs[0]
    B = len(x0)
    for (it1, n1, e1, x1, v1, c1, mc1) in runs[1:]:
        assert B == blocks and it0 == it1 and n0 == n1 == it0 * (3 * B - 2)
        assert e0 == 0 and e1 == it1 * 2 * (B - 1)
        assert abs(c0 - c1) < 1e-7 * c0 and abs(mc0 - mc1) < 1e-9
        for b in range(B):
            assert np.abs(x0[b] - x1[b]).max() < 1e-8
            assert np.abs(v0[b] - v1[b]).max() < 1e-8 * np.abs(v0[b]).max()


@pytest.mark.parametrize("mt,blocks,terr,condensed", [(True, 6, False, True), (False, 4, False, True), (True, 4, True, True), (False, 5, False, False)])
def test_information_form_of_the_junction_carry(built, orc, tmp_path, mt, blocks, terr, condensed):
    """dnagpu_schur_carry in information form (default: the junction's weight matrix S and reduced right-hand side r travel, the
    receiving block adds r + S (the sender's estimates - its own); S is never inverted) against its estimates form
    (dnagpu_debug_set_info_carry(0): S inverted, estimates + S^-1 r carried, CarryStnEstimatesandVariancesForward dnaadjust.cpp:998-1128):
    same adjustment, every iteration's correction, every estimate and variance; fewer flops counted.  A terrestrial network has its
    estimates move between the iterations (the linearisation point of both blocks of a junction is the same one)."""
    if terr:
        from tests import terrestrial_net as T
        T.build_mixed_network(str(tmp_path / "c"), 6, 4, blocks, seed=5)
    else:
        adjust.write_synthetic_network(str(tmp_path), "c", 18, 12, 0, blocks, seed=8, x_clusters=10, y_cluster=True, initial_sigma=0.3)
    runs = []
    for form in (0, 1):
        old = built.dnagpu_debug_set_info_carry(form)
        try:
            a, st = _device_run(str(tmp_path), "c", True, multi_thread=mt, keep_factors=condensed, defer_variances=2 if condensed else 0)
        finally:
            built.dnagpu_debug_set_info_carry(old)
        assert st == 0 and a.elimination_count() > 0
        a.GenerateStatistics()
        runs.append((a.CurrentIteration(), [a.block_estimates(b) for b in range(a.blockCount())], [a.block_variances_packed(b) for b in range(a.blockCount())],
                     a.GetChiSquared(), [a.GetIterationCorrection(i + 1) for i in range(a.CurrentIteration())], a.algorithmic_flops()))
        if form == 1:
            net = orc.Network(str(tmp_path / "c"), True)
            o = orc.Adjustment(net, True)
            o.prepare()
            _compare(a, st, o, o.run())
            o.close()
        a.close()
    (it0, x0, v0, c0, corr0, f0), (it1, x1, v1, c1, corr1, f1) = runs
    assert it0 == it1 and f1 < f0
    assert abs(c0 - c1) < 1e-8 * c0 and np.abs(np.array(corr0) - np.array(corr1)).max() < 1e-9
    for b in range(len(x0)):
        assert np.abs(x0[b] - x1[b]).max() < 1e-9
        assert np.abs(v0[b] - v1[b]).max() < 1e-9 * np.abs(v0[b]).max()


@pytest.mark.parametrize("mt", [False, True])
def test_condensed_reuse_across_iterations(built, orc, tmp_path, mt):
    """a.reuse_inverses with the condensed schedule and kept factors: from the second iteration on a block is neither condensed
    nor inverted again -- reduced right-hand sides from the kept factor, chains on the condensed blocks, products with the
    resident rigorous variances.  Same results (the matrices of a GNSS-only network do not change between iterations)."""
    adjust.write_synthetic_network(str(tmp_path), "r", 14, 12, 0, 5, seed=9, x_clusters=20, y_cluster=True, initial_sigma=0.4)
    runs = []
    for reuse in (False, True):
        a, st = _device_run(str(tmp_path), "r", True, multi_thread=mt, reuse_inverses=reuse)
        assert st == 0 and a.CurrentIteration() >= 2
        a.GenerateStatistics()
        runs.append((a.CurrentIteration(), a.completion_count(), [a.block_estimates(b) for b in range(a.blockCount())],
                     [a.block_variances_packed(b) for b in range(a.blockCount())], a.GetChiSquared(),
                     [a.GetIterationCorrection(i + 1) for i in range(a.CurrentIteration())]))
        a.close()
    (it0, n0, x0, v0, c0, corr0), (it1, n1, x1, v1, c1, corr1) = runs
    B = len(x0)
    assert it0 == it1 and n0 == it0 * B and n1 == B            # one completion per block in all, not per iteration
    _against_oracle(orc, str(tmp_path / "r"), it1, x1, v1, corr1)
    assert abs(c0 - c1) < 1e-9 * c0 and np.abs(np.array(corr0) - np.array(corr1)).max() < 1e-10
    for b in range(B):
        assert np.abs(x0[b] - x1[b]).max() < 1e-9
        assert np.array_equal(v0[b], v1[b])                     # the very inverse of iteration 1


@pytest.mark.parametrize("mt,blocks", [(False, 5), (True, 6)])
def test_factor_reuse_across_iterations(built, orc, tmp_path, mt, blocks):
    """a.reuse_factors (default): in a GNSS-only network the normals of an iteration do not depend on the estimates -- the reference's own
    test in simultaneous mode, SolveTry(CurrentIteration() < 2 || ContainsNonGPS()), dnaadjust.cpp:2452-2457 -- so from the second iteration
    on every block reduces and solves its right-hand side with the light factor of iteration 1, the chain steps on the condensed blocks take
    theirs through their kept factors, and nothing is factored again.  Same iterations, corrections and estimates to rounding; the variance
    matrices come from the same factor bits either way."""
    adjust.write_synthetic_network(str(tmp_path), "r", 14, 12, 0, blocks, seed=9, x_clusters=20, y_cluster=True, initial_sigma=0.4)
    runs = []
    for reuse in (False, True):
        a, st = _device_run(str(tmp_path), "r", True, multi_thread=mt, reuse_factors=reuse)
        assert st == 0 and a.CurrentIteration() >= 2
        a.GenerateStatistics()
        runs.append((a.CurrentIteration(), a.factor_reuses(), a.chain_step_reuses(), a.algorithmic_flops(),
                     [a.block_estimates(b) for b in range(a.blockCount())], [a.block_variances_packed(b) for b in range(a.blockCount())],
                     a.GetChiSquared(), [a.GetIterationCorrection(i + 1) for i in range(a.CurrentIteration())]))
        if reuse:
            # a second adjustment on the same object starts over: the kept factors are those of ONE adjustment
            a.ResetAdjustment()
            assert a.AdjustNetwork() == 0 and a.CurrentIteration() == runs[-1][0]
            assert a.factor_reuses() == runs[-1][1] and a.chain_step_reuses() == runs[-1][2]
            for b in range(a.blockCount()):
                assert np.array_equal(a.block_estimates(b), runs[-1][4][b])
        a.close()
    (it0, f0, c0, fl0, x0, v0, chi0, corr0), (it1, f1, c1, fl1, x1, v1, chi1, corr1) = runs
    B = len(x0)
    assert it0 == it1 and f0 == 0 and c0 == 0
    assert f1 == (it1 - 1) * B                       # every block keeps its factor on this small network
    assert c1 == (it1 - 1) * (2 * B - 4)             # ... and every chain step that eliminates anything (the two end steps do not)
    assert fl1 < fl0                                 # one round of factorisations + the variance matrices, not it1 rounds
    _against_oracle(orc, str(tmp_path / "r"), it1, x1, v1, corr1)
    assert abs(chi0 - chi1) < 1e-9 * chi0 and np.abs(np.array(corr0) - np.array(corr1)).max() < 1e-10
    for b in range(B):
        assert np.abs(x0[b] - x1[b]).max() < 1e-9
        assert np.abs(v0[b] - v1[b]).max() <= 1e-13 * np.abs(v0[b]).max()


def test_a_singular_chain_step_among_many_is_named(built, tmp_path):
    """40 small blocks: the chains take the elimination's verdict once per chain, not per step (dnagpu_chain_hold_info); a step that
    meets a pivot that is not positive makes the phase run again step by step, and the reference's message names the block
    (dnamatrix_contiguous.cpp:983 through SolveTry, dnaadjust.cpp:6575-6582)"""
    adjust.write_synthetic_network(str(tmp_path), "s", 80, 24, 0, 1, seed=12, rows_lo=2, rows_hi=2)
    base = os.path.join(str(tmp_path), "s")
    ISL, JSL, CML, nets = F.read_seg(base + ".seg")
    assert len(ISL) == 40
    msr = F.read_bms(base + ".bms")
    target = int(JSL[20][len(JSL[20]) // 2])       # a junction station loses every measurement; free stations weigh nothing: a zero pivot
    for k in range(len(CML)):
        CML[k] = np.array([int(i) for i in CML[k] if int(msr[int(i)]["station1"]) != target and int(msr[int(i)]["station2"]) != target], dtype=np.uint32)
    F.write_seg(base + ".seg", ISL, JSL, CML, nets, msr)
    for mt in (False, True):
        with pytest.raises(adjust.NetAdjustException) as e:
            _device_run(str(tmp_path), "s", True, multi_thread=mt, free_std_dev=1e200)
        import re
        assert "singular" in str(e.value) and re.search(r"block (19|20|21|22)\b", str(e.value)), str(e.value)


@pytest.mark.parametrize("mt", [False, True])
def test_many_small_blocks_in_one_launch(built, orc, tmp_path, mt):
    """a dnasegment-like cut into 40 small blocks (strips of 2 rows of 24 stations; include/config/dnaoptions.hpp:382 makes 150-station blocks
    by default): the rigorous solve of ALL blocks is one launch in every iteration, from iteration 2 on (a.reuse_factors) the condensing step as well
    (dnagpu_small_batch_*: a workgroup per block), the chain steps one launch per step (dnagpu_chain_step_rhs).  Against the oracle, and
    against the run in which every iteration factors again."""
    info = adjust.write_synthetic_network(str(tmp_path), "m", 80, 24, 0, 1, seed=12, rows_lo=2, rows_hi=2, initial_sigma=0.3)
    assert info["blocks"] == 40
    net = orc.Network(str(tmp_path / "m"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "m", True, multi_thread=mt)
    _compare(a, st, o, ost)
    it = a.CurrentIteration()
    assert it >= 2 and a.small_batch_steps() == it * 40 and a.factor_reuses() == (it - 1) * 40      # (iteration 1: the rigorous solves only)
    assert a.chain_step_reuses() == (it - 1) * (2 * 40 - 4)
    fd, rec = _compare_statistics(a, o)
    x1 = [a.block_estimates(b) for b in range(40)]
    v1 = [a.block_variances_packed(b) for b in range(40)]
    # the same handle again (the table of the blocks is kept), and the run without reuse
    a.ResetAdjustment()
    assert a.AdjustNetwork() == st and a.small_batch_steps() == it * 40
    for b in range(40):
        assert np.array_equal(a.block_estimates(b), x1[b])
    a.close()
    r, st2 = _device_run(str(tmp_path), "m", True, multi_thread=mt, reuse_factors=False)
    assert st2 == st and r.CurrentIteration() == it and r.small_batch_steps() == 0
    for b in range(40):
        assert np.abs(r.block_estimates(b) - x1[b]).max() < 1e-9
        assert np.abs(r.block_variances_packed(b) - v1[b]).max() <= 1e-12 * np.abs(v1[b]).max()
    r.close()
    o.close()


@pytest.mark.parametrize("mt,runs,reuse", [(True, 4, True), (False, 7, True), (True, 13, False), (True, -1, True)])
def test_chains_in_lock_step(built, orc, tmp_path, mt, runs, reuse):
    """a.chain_runs: the two junction chains of a many-block network cut into runs whose steps advance together (dna_adjust::LockstepChains:
    every run merged to its end stations, the chains over the runs, the chains inside every run from the boundary values -- each level's
    steps of all runs as batches of merged launches, dnagpu_chain_plan_*).  Against the oracle and against the chains step by step
    (same additions in the same order inside a run: agreement to rounding); runs of equal and of unequal length, more runs than a batch
    holds would need (13 runs of 3 - 4 blocks), one chain and two, with and without factor reuse, twice on one handle."""
    nb = 70 if runs < 0 else 40
    info = adjust.write_synthetic_network(str(tmp_path), "m", 2 * nb, 24, 0, 1, seed=12, rows_lo=2, rows_hi=2, initial_sigma=0.3)
    assert info["blocks"] == nb
    net = orc.Network(str(tmp_path / "m"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "m", True, multi_thread=mt, chain_runs=runs, reuse_factors=reuse)
    assert a.chain_runs() == (16 if runs < 0 else runs)
    _compare(a, st, o, ost)
    it = a.CurrentIteration()
    assert it >= 2 and a.chain_step_reuses() == ((it - 1) * (2 * nb - 2) if reuse else 0)
    _compare_statistics(a, o)
    x1 = [a.block_estimates(b) for b in range(nb)]
    v1 = [a.block_variances_packed(b) for b in range(nb)]
    a.ResetAdjustment()
    assert a.AdjustNetwork() == st and a.CurrentIteration() == it
    for b in range(nb):
        assert np.array_equal(a.block_estimates(b), x1[b])
    a.close()
    r, st2 = _device_run(str(tmp_path), "m", True, multi_thread=mt, chain_runs=0, reuse_factors=reuse)
    assert r.chain_runs() == 0 and st2 == st and r.CurrentIteration() == it
    for b in range(nb):
        assert np.abs(r.block_estimates(b) - x1[b]).max() < 1e-9
        assert np.abs(r.block_variances_packed(b) - v1[b]).max() <= 1e-11 * np.abs(v1[b]).max()
    r.close()
    o.close()


@pytest.mark.parametrize("seed", range(8))
def test_lock_step_chains_on_random_segmentations(built, orc, tmp_path, seed):
    """the lock-step chains on cuts the strip generator cannot make (tests/segfuzz.py): junction stations that stay junction over several
    blocks -- they persist through the merges of a run and across run boundaries --, junction sets of uneven size (the members of a batch
    are eliminated in the largest member's padded shape), runs of unequal length.  Oracle live."""
    from tests import segfuzz
    rng = np.random.default_rng(4000 + seed)
    adjust.write_synthetic_network(str(tmp_path), "a", int(rng.integers(16, 26)), int(rng.integers(5, 9)), 0, 1, seed=seed * 5 + 1, initial_sigma=0.2)
    info = segfuzz.write_cut(str(tmp_path / "a"), rng, mean_block=int(rng.integers(4, 9)), noise=float(rng.uniform(0.1, 0.5)), lone_last=False)
    assert info["blocks"] >= 9 and info["nets"] == 1
    net = orc.Network(str(tmp_path / "a"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    assert ost == 0
    runs = 2 + seed % 3
    a, st = _device_run(str(tmp_path), "a", True, multi_thread=bool(seed % 2), chain_runs=runs)
    assert a.chain_runs() == runs, info
    _compare(a, st, o, ost)
    a.close()
    o.close()


def test_lock_step_chains_without_room_for_their_factors(built, orc, tmp_path, monkeypatch):
    """the steps' factors of a chain plan are kept while they fit what PrepareAdjustment sets aside for chain steps' factors; beyond that the plan
    keeps none and every iteration eliminates again -- in lock step all the same (DNAGPU_FACTOR_BUDGET_GB: the memory-tight plan at any size)"""
    info = adjust.write_synthetic_network(str(tmp_path), "m", 80, 24, 0, 1, seed=12, rows_lo=2, rows_hi=2, initial_sigma=0.3)
    net = orc.Network(str(tmp_path / "m"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    monkeypatch.setenv("DNAGPU_FACTOR_BUDGET_GB", "0.0002")
    a, st = _device_run(str(tmp_path), "m", True, multi_thread=True, chain_runs=4)
    assert a.chain_runs() == 4 and a.CurrentIteration() >= 2 and a.chain_step_reuses() == 0
    _compare(a, st, o, ost)
    a.close()
    o.close()


@pytest.mark.parametrize("seed", range(6))
def test_lock_step_chains_over_several_networks(built, orc, tmp_path, seed):
    """a project with three network ids -- two randomly cut contiguous networks and an isolated block (tests/test_oracle_adjust.py::_fuzzed_project):
    the chains of the networks are independent of each other and advance together like the runs of one (the runs are dealt to the networks by
    their length; a short network is one run); the isolated block has no chain step.  Oracle live."""
    from tests.test_oracle_adjust import _fuzzed_project
    info = _fuzzed_project(tmp_path, 50 + seed)
    net = orc.Network(str(tmp_path / "all"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    assert ost == 0
    a, st = _device_run(str(tmp_path), "all", True, multi_thread=bool(seed % 2), chain_runs=3 + seed % 2)
    assert a.chain_runs() >= 2, info
    _compare(a, st, o, ost)
    a.close()
    o.close()


def test_a_singular_step_of_lock_step_chains_is_named(built, tmp_path):
    """the lock-step chains take the eliminations' verdict once per level; a pivot that is not positive sends the phase to the chains step
    by step, which name the block (test_a_singular_chain_step_among_many_is_named)"""
    adjust.write_synthetic_network(str(tmp_path), "s", 80, 24, 0, 1, seed=12, rows_lo=2, rows_hi=2)
    base = os.path.join(str(tmp_path), "s")
    ISL, JSL, CML, nets = F.read_seg(base + ".seg")
    msr = F.read_bms(base + ".bms")
    target = int(JSL[20][len(JSL[20]) // 2])
    for k in range(len(CML)):
        CML[k] = np.array([int(i) for i in CML[k] if int(msr[int(i)]["station1"]) != target and int(msr[int(i)]["station2"]) != target], dtype=np.uint32)
    F.write_seg(base + ".seg", ISL, JSL, CML, nets, msr)
    with pytest.raises(adjust.NetAdjustException) as e:
        _device_run(str(tmp_path), "s", True, multi_thread=True, free_std_dev=1e200, chain_runs=5)
    import re
    assert "singular" in str(e.value) and re.search(r"block (19|20|21|22)\b", str(e.value)), str(e.value)


def test_phased_block_1_mode(built, orc, tmp_path):
    """Phased_Block_1Mode (AdjustPhasedBlock1, dnaadjust.cpp:2675): one reverse pass; block 1 is rigorous -- exactly what the first
    iteration of the full phased adjustment gives it -- the blocks between keep their reverse solution, the last block is not
    finalised at all (UpdateEstimatesFinal returns for it, ADJ:3750).  Against the oracle's restatement of the mode, block by block."""
    adjust.write_synthetic_network(str(tmp_path), "b", 24, 10, 0, 4, seed=6, x_clusters=8)
    net = orc.Network(str(tmp_path / "b"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run_block1()
    p = adjust.ProjectSettings("b", str(tmp_path), adjust_mode=adjust.Phased_Block_1Mode)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    st = a.AdjustNetwork()
    assert a.CurrentIteration() == 1 and st == ost and st in (adjust.ADJUST_SUCCESS, adjust.ADJUST_THRESHOLD_EXCEEDED)
    assert abs(a.GetMaxCorrection() - o.max_correction(1)) < 1e-8 and abs(a.GetMaxCorrection()) > 0.0
    last = a.blockCount() - 1
    for b in range(a.blockCount()):
        assert np.abs(a.block_estimates(b) - o.block_estimates(b)).max() < TOL_X, b
        if b != last:
            vo = o.block_variances(b)
            assert np.abs(a.block_variances_packed(b) - vo).max() / np.abs(vo).max() < TOL_V, b
    # block 1 equals the first iteration of the full phased adjustment (reference schedule) bit for bit
    f, st_f = _device_run(str(tmp_path), "b", True, max_iterations=1, schur_carry=False)
    assert np.array_equal(a.block_estimates(0), f.block_estimates(0))
    assert np.array_equal(a.block_variances_packed(0), f.block_variances_packed(0))
    # a block in between has only seen the blocks after it: it differs from the rigorous solution
    assert np.abs(a.block_estimates(last - 1) - f.block_estimates(last - 1)).max() > 1e-6
    a.close()
    f.close()
    o.close()


def test_deserialise_adjusted_variance_matrices(built, tmp_path):
    """-rva.mtx / -pam.mtx written by one adjustment and read back by another dna_adjust (dnaadjust --report-results,
    DeSerialiseAdjustedVarianceMatrices dnaadjust.cpp:6720)"""
    adjust.write_synthetic_network(str(tmp_path), "s", 14, 9, 0, 3, seed=12, x_clusters=6)
    a, st = _device_run(str(tmp_path), "s", True)
    a.GenerateStatistics()
    a.SerialiseAdjustedVarianceMatrices()
    v = [a.block_variances_packed(b) for b in range(a.blockCount())]
    p = [a.block_prec_adj_msrs(b) for b in range(a.blockCount())]
    a.close()
    b = adjust.DnaAdjust()
    b.PrepareAdjustment(adjust.ProjectSettings("s", str(tmp_path), adjust_mode=adjust.PhasedMode))
    b.DeSerialiseAdjustedVarianceMatrices()
    for k in range(b.blockCount()):
        assert np.array_equal(b.block_variances_packed(k), v[k])
        assert np.array_equal(b.block_prec_adj_msrs(k), p[k])
    b.close()
    # files of another network are refused
    adjust.write_synthetic_network(str(tmp_path), "t", 10, 9, 0, 3, seed=12)
    os.replace(str(tmp_path / "s-rva.mtx"), str(tmp_path / "t-rva.mtx"))
    os.replace(str(tmp_path / "s-pam.mtx"), str(tmp_path / "t-pam.mtx"))
    c = adjust.DnaAdjust()
    c.PrepareAdjustment(adjust.ProjectSettings("t", str(tmp_path), adjust_mode=adjust.PhasedMode))
    with pytest.raises(adjust.NetAdjustException) as e:
        c.DeSerialiseAdjustedVarianceMatrices()
    assert "does not match the dimensions" in str(e.value)
    c.close()


@pytest.mark.parametrize("mt", [False, True])
def test_staged_rigorous_variances_in_host_memory(built, orc, tmp_path, mt):
    """a.stage: the rigorous variance matrices leave HBM for page-locked host memory (what the reference's staged adjustment does
    with memory-mapped files); identical results, statistics and result files included"""
    adjust.write_synthetic_network(str(tmp_path), "g", 16, 10, 0, 4, seed=21, x_clusters=10, y_cluster=True)
    runs = []
    for stage in (False, True):
        a, st = _device_run(str(tmp_path), "g", True, multi_thread=mt, stage=stage, output_folder=str(tmp_path))
        assert st == 0 and bool(a.lib.dnaadj_staged(a.h)) == stage
        a.GenerateStatistics()
        a.SerialiseAdjustedVarianceMatrices()
        runs.append(([a.block_estimates(b) for b in range(a.blockCount())], [a.block_variances_packed(b) for b in range(a.blockCount())],
                     a.GetChiSquared(), a.GetGlobalPelzerRel(), np.frombuffer(a.measurement_records().tobytes(), dtype=np.uint8).copy(),
                     open(str(tmp_path / "g-rva.mtx"), "rb").read(), open(str(tmp_path / "g-pam.mtx"), "rb").read()))
        a.close()
    (x0, v0, c0, p0, r0, f0, g0), (x1, v1, c1, p1, r1, f1, g1) = runs
    for b in range(len(x0)):
        assert np.array_equal(x0[b], x1[b]) and np.array_equal(v0[b], v1[b])
    assert c0 == c1 and p0 == p1 and np.array_equal(r0, r1) and f0 == f1 and g0 == g1


@pytest.mark.parametrize("host_gb", ["0", "0.0002"])
def test_staged_store_past_the_host_memory_limit(built, tmp_path, monkeypatch, host_gb):
    """the staged store where the host's memory limit ends (DecideStaging: the container's cgroup, here DNAGPU_HOST_STORE_GB): the blocks
    past it keep their packed variance matrix in HBM instead (block_t::rig_on_device).  "0": every block; 200 kB: the first two blocks in
    host memory, the rest in HBM.  Identical results -- estimates, variances, statistics, the re-read result files -- and the plan
    says where the bytes are."""
    adjust.write_synthetic_network(str(tmp_path), "g", 16, 10, 0, 4, seed=21, x_clusters=10, y_cluster=True)
    runs = []
    for limit in (None, host_gb):
        if limit is None:
            monkeypatch.delenv("DNAGPU_HOST_STORE_GB", raising=False)
        else:
            monkeypatch.setenv("DNAGPU_HOST_STORE_GB", limit)
        a, st = _device_run(str(tmp_path), "g", True, multi_thread=True, stage=True, output_folder=str(tmp_path))
        assert st == 0 and a.lib.dnaadj_staged(a.h)
        plan = a.memory_plan()
        if limit is None:
            assert plan["staged_variances_packed_in_hbm_bytes"] == 0 and plan["staged_variances_host_bytes"] > 0
        elif limit == "0":
            assert plan["staged_variances_host_bytes"] == 0 and plan["staged_variances_packed_in_hbm_bytes"] > 0
        else:
            assert plan["staged_variances_host_bytes"] > 0 and plan["staged_variances_packed_in_hbm_bytes"] > 0
        a.GenerateStatistics()
        a.SerialiseAdjustedVarianceMatrices()
        files = (open(str(tmp_path / "g-rva.mtx"), "rb").read(), open(str(tmp_path / "g-pam.mtx"), "rb").read())
        first = ([a.block_estimates(b) for b in range(a.blockCount())], [a.block_variances_packed(b) for b in range(a.blockCount())],
                 a.GetChiSquared(), np.frombuffer(a.measurement_records().tobytes(), dtype=np.uint8).copy(), files)
        a.DeSerialiseAdjustedVarianceMatrices()          # (back into the same store)
        again = [a.block_variances_packed(b) for b in range(a.blockCount())]
        for v, w in zip(first[1], again):
            assert np.array_equal(v, w)
        runs.append(first)
        a.close()
    (x0, v0, c0, r0, f0), (x1, v1, c1, r1, f1) = runs
    for b in range(len(x0)):
        assert np.array_equal(x0[b], x1[b]) and np.array_equal(v0[b], v1[b])
    assert c0 == c1 and np.array_equal(r0, r1) and f0 == f1


def test_bench_contract_small_workload(built):
    """bench.py end to end on the smoke-size workload: ONE JSON line with the contract's keys, roofline and check blocks"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "small", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "check"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "stations/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - d["config"]["stations"] / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0.5 < d["check"]["sigma_zero"] < 1.5 and d["check"]["max_abs_error_vs_truth_m"] < 0.2


@pytest.mark.parametrize("extra", [[], ["--reference-schedule"]])
def test_bench_distributed_path_over_rccl(built, extra):
    """bench.py's N > 1 path with the NCCL (= RCCL) backend and device-resident payloads, forced onto the one rank a 1-GPU
    box has: process group on cuda:0, condensed / junction payloads exported into device tensors, broadcast, all_reduce and
    all_gather through RCCL, the JSON of the distributed leg -- same answer as the single-process bench"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DNAGPU_FORCE_DISTRIBUTED="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = []
    for e in (env, dict(os.environ)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "small", "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600, cwd=root, env=e)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        out.append(json.loads(lines[0]))
    d, s = out
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["config"]["stations"] == s["config"]["stations"]
    assert d["config"]["iterations_to_converge"] == s["config"]["iterations_to_converge"]
    assert abs(d["check"]["sigma_zero"] - s["check"]["sigma_zero"]) < 1e-9


@pytest.mark.parametrize("blocks,mt", [(1, False), (3, False), (4, True)])
def test_mixed_station_constraints_on_the_device(built, orc, tmp_path, blocks, mt):
    """CCF / CFF / FFC / CFC / FCC station constraints on geographic, projection and cartesian station records
    (FormConstraintStationVarianceMatrix, dnaadjust.cpp:2041-2137): facade against oracle (itself checked against numpy)"""
    adjust.write_synthetic_network(str(tmp_path), "m", 9, 6, 0, blocks, seed=31, x_clusters=6)
    base = str(tmp_path / "m")
    bst = F.read_bst(base + ".bst").copy()
    for s, (code, typ) in {3: (b"CCF", 2), 8: (b"FFC", 1), 14: (b"CFF", 3), 20: (b"CFC", 3), 27: (b"FCC", 0), 33: (b"FFC", 0), 50: (b"CCF", 3)}.items():
        bst["stationConst"][s] = code
        bst["suppliedStationType"][s] = typ
    F.write_bst(base + ".bst", bst)
    phased = blocks > 1
    net = orc.Network(base, phased)
    o = orc.Adjustment(net, phased)
    o.prepare()
    ost = o.run()
    a, st = _device_run(str(tmp_path), "m", phased, multi_thread=mt)
    _compare(a, st, o, ost)
    assert a.GetUnknownsCount() == 3 * 54 - sum(c.count(b"C") for c in [bytes(x[:3]) for x in bst["stationConst"]])
    a.close()
    o.close()


def test_multi_chain_runs_are_reproducible(built, tmp_path):
    """four chains, host threads taking blocks from a queue: whichever chain serves a block, the bits are the same -- three runs
    of the same adjustment (and the one-chain run) give identical coordinates, variances and statistics"""
    adjust.write_synthetic_network(str(tmp_path), "q", 40, 16, 0, 8, seed=17, x_clusters=10, y_cluster=True)
    ref = None
    for mt in (True, True, True, False):
        a, st = _device_run(str(tmp_path), "q", True, multi_thread=mt)
        assert st == 0
        a.GenerateStatistics()
        cur = ([a.block_estimates(b).tobytes() for b in range(a.blockCount())], [a.block_variances_packed(b).tobytes() for b in range(a.blockCount())],
               a.GetChiSquared(), a.GetGlobalPelzerRel())
        a.close()
        if ref is None:
            ref = cur
        assert cur == ref


@pytest.mark.parametrize("mt", [False, True])
def test_cancel_adjustment(built, tmp_path, mt):
    """CancelAdjustment() (the reference's SIGINT handler, dnaadjustwrapper.cpp): an adjustment cancelled before its first block
    stops with ADJUST_CANCELLED; one cancelled from another thread while it runs stops at the next block (or had already
    finished); the object adjusts normally afterwards"""
    import threading
    import time
    adjust.write_synthetic_network(str(tmp_path), "k", 120, 40, 0, 12, seed=5)
    p = adjust.ProjectSettings("k", str(tmp_path), adjust_mode=adjust.PhasedMode, multi_thread=mt)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    a.CancelAdjustment()
    assert a.AdjustNetwork() == adjust.ADJUST_CANCELLED
    a.ResetAdjustment()
    out = {}
    t = threading.Thread(target=lambda: out.setdefault("st", a.AdjustNetwork()))
    t.start()
    time.sleep(0.002)
    a.CancelAdjustment()
    t.join(timeout=120)
    assert not t.is_alive() and out["st"] in (adjust.ADJUST_CANCELLED, adjust.ADJUST_SUCCESS)
    a.ResetAdjustment()
    assert a.AdjustNetwork() == adjust.ADJUST_SUCCESS
    a.close()
