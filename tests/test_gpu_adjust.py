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


def test_iteration_limit_and_threshold(built, orc, golden_dir):
    a, st = _device_run(golden_dir, "tiny_net", True, max_iterations=1)
    assert st == adjust.ADJUST_MAX_ITERATIONS_EXCEEDED and a.CurrentIteration() == 1
    a.close()
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


def test_orchestrator_single_rank_equals_facade(built, tmp_path):
    """dynadjust_amd/parallel.run_phased on one rank drives the same per-block steps as AdjustPhased"""
    from dynadjust_amd import parallel
    import torch
    adjust.write_synthetic_network(str(tmp_path), "n", 30, 12, 0, 5, seed=9)
    f, st_f = _device_run(str(tmp_path), "n", True)
    p = adjust.ProjectSettings("n", str(tmp_path), adjust_mode=adjust.PhasedMode)
    be = parallel.DeviceBlockBackend(p, torch.device("cpu"))
    st, its, corr = parallel.run_phased(be, None, 0, 1)
    assert st == st_f and its == f.CurrentIteration()
    for b in range(f.blockCount()):
        assert np.array_equal(be.adj.block_estimates(b), f.block_estimates(b))
        assert np.array_equal(be.adj.block_variances_packed(b), f.block_variances_packed(b))
    be.close()
    f.close()


def _two_rank_worker(rank, world, port, folder, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dynadjust_amd import adjust as adj, parallel
    p = adj.ProjectSettings("n", folder, adjust_mode=adj.PhasedMode)
    be = parallel.DeviceBlockBackend(p, torch.device("cpu"))     # both ranks share the box's single GPU; payloads via host
    st, its, corr = parallel.run_phased(be, dist, rank, world)
    sch = parallel.PhasedSchedule([be.flags(k) for k in range(be.n_blocks)], world)
    res = {"status": st, "iterations": its}
    for k in range(be.n_blocks):
        res[f"coords_{k}"] = be.get_coords(k)
        if sch.final_owner(k) == rank:
            res[f"var_{k}"] = be.adj.block_variances_packed(k)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **res)
    dist.barrier()
    be.close()
    dist.destroy_process_group()


def test_orchestrator_two_ranks_device_backend(built, orc, tmp_path):
    """the real device backend under a 2-rank schedule (gloo transport, both processes on this box's GPU):
    junction export/import, combination solves on the 'other' rank, coordinate all_reduce"""
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
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path), str(tmp_path)), nprocs=2, join=True)
    seen = set()
    for r in range(2):
        res = np.load(str(tmp_path / f"rank{r}.npz"))
        assert int(res["status"]) == ost and int(res["iterations"]) == o.iterations()
        for k in range(6):
            assert np.abs(res[f"coords_{k}"] - o.block_estimates(k)).max() < TOL_X
            if f"var_{k}" in res:
                seen.add(k)
                vo = o.block_variances(k)
                assert np.abs(res[f"var_{k}"] - vo).max() / np.abs(vo).max() < TOL_V
    assert seen == set(range(6))
    o.close()
