"""N ranks on N GPUs over RCCL / xGMI: the tests a one-GPU box has to skip (tests/test_gpu_distributed.py runs the same driver there
with ranks that share the device).  Every test needs `torch.cuda.device_count() >= N` and skips otherwise.

  * cfg3 (BASELINE.json configs[2]) on 2 / 4 / 8 GPUs through the C++ driver (a.devices: one process, RCCL communicators made by the
    library's per-GPU threads) against the committed record of the CPU oracle's run, tests/golden/cfg3_oracle.npz;
  * a small network, both schedules, over RCCL against the oracle run live; statistics and result files against the one-GPU run;
  * the inverse of ONE block split over 2 GPUs (simultaneous mode), against the one-GPU inverse;
  * one process per GPU with nothing shared but MASTER_ADDR / MASTER_PORT: the TCP hand-off of the ncclUniqueId.
Replaces the thread pool of dna_adjust::AdjustPhasedMultiThread (dnaadjust-multi.cpp:92-244) across devices."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from dynadjust_amd import adjust
from tests import dnaformats as F
from tests import fullsize

pytestmark = pytest.mark.gpu

TOL_X = 1e-8
TOL_V = 1e-8


def _need(n):
    from dynadjust_amd import _lib
    have = _lib.load().dnagpu_device_count()      # the library's own count: importing torch here costs minutes on a cold box
    if have < n:
        pytest.skip(f"needs {n} GPUs, this node shows {have}")


def _run(folder, name, mode=adjust.PhasedMode, **kw):
    p = adjust.ProjectSettings(name, folder, adjust_mode=mode, **kw)
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(p)
    return a


@pytest.mark.parametrize("ngpu", [2, 4, 8])
def test_cfg3_over_rccl_against_the_oracle_record(built, golden_dir, tmp_path, ngpu):
    _need(ngpu)
    path = os.path.join(golden_dir, "cfg3_oracle.npz")
    g = np.load(path)
    meta = json.loads(bytes(g["meta"]).decode())
    rows, cols, nbl, blocks, phased = fullsize.WORKLOADS["cfg3"]
    info = adjust.write_synthetic_network(str(tmp_path), "net", rows, cols, nbl, blocks, seed=fullsize.SEED)
    assert info["stations"] == meta["stations"]
    a = _run(str(tmp_path), "net", devices=list(range(ngpu)), multi_thread=True)
    st = a.AdjustNetworkDistributed()
    rank, world, transport = a.dist_info()
    assert (rank, world, transport) == (0, ngpu, "rccl")
    assert a.device_instance_stats(0)["rccl_ranks"] == ngpu
    assert st == meta["status"] and a.CurrentIteration() == meta["iterations"]
    for i, c in enumerate(meta["corrections"]):
        assert abs(a.GetIterationCorrection(i + 1) - c) < TOL_X
    owners = [a.block_owner(k) for k in range(a.blockCount())]
    assert owners == sorted(owners) and sorted(set(owners)) == list(range(ngpu))
    dx = dv = 0.0
    for b in range(a.blockCount()):
        assert np.array_equal(a.block_stations(b), g[f"stations_{b}"])
        est = a.block_estimates(b)
        dx = max(dx, float(np.abs(est - g[f"estimates_{b}"]).max()))
        diag, vcols = fullsize.sample_packed(a.block_variances_packed(b), est.size)       # (fetched from the GPU that holds it)
        scale = float(np.abs(g[f"vdiag_{b}"]).max())
        dv = max(dv, float(np.abs(diag - g[f"vdiag_{b}"]).max()) / scale, float(np.abs(vcols - g[f"vcols_{b}"]).max()) / scale)
    a.GenerateStatistics()
    assert dx < TOL_X and dv < TOL_V, (dx, dv)
    assert a.GetDegreesOfFreedom() == meta["dof"]
    assert abs(a.GetChiSquared() - meta["chi_squared"]) / meta["chi_squared"] < 1e-7
    ex = [a.device_instance_stats(r) for r in range(ngpu)]
    assert all(e["exchanged_bytes"] > 0 for e in ex)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump({"workload": "cfg3", "gpus": ngpu, "transport": transport, "max_abs_dx_m": dx, "max_rel_dvar": dv,
                   "blocks_per_rank": [owners.count(r) for r in range(ngpu)],
                   "exchanged_bytes_per_rank": [e["exchanged_bytes"] for e in ex]}, open(os.path.join(out, f"parity_cfg3_{ngpu}gpus.json"), "w"), indent=1)
    except OSError:
        pass
    a.close()


@pytest.mark.parametrize("ngpu,schur,two_level", [(2, True, True), (2, False, True), (3, True, False), (4, True, True)])
def test_small_network_over_rccl(built, orc, tmp_path, ngpu, schur, two_level):
    _need(ngpu)
    adjust.write_synthetic_network(str(tmp_path), "n", 40, 12, 0, 8, seed=10)
    net = orc.Network(str(tmp_path / "n"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    os.makedirs(tmp_path / "multi", exist_ok=True)
    os.makedirs(tmp_path / "single", exist_ok=True)
    a = _run(str(tmp_path), "n", devices=list(range(ngpu)), schur_carry=schur, dist_two_level=two_level, output_folder=str(tmp_path / "multi"))
    st = a.AdjustNetworkDistributed()
    assert a.dist_info() == (0, ngpu, "rccl")
    assert st == ost and a.CurrentIteration() == o.iterations()
    for k in range(8):
        assert np.abs(a.block_estimates(k) - o.block_estimates(k)).max() < TOL_X
        vo = o.block_variances(k)
        assert np.abs(a.block_variances_packed(k) - vo).max() / np.abs(vo).max() < TOL_V
    a.GenerateStatistics()
    a.SerialiseAdjustedVarianceMatrices()
    f = _run(str(tmp_path), "n", schur_carry=schur, output_folder=str(tmp_path / "single"))
    assert f.AdjustNetwork() == st
    f.GenerateStatistics()
    f.SerialiseAdjustedVarianceMatrices()
    assert abs(a.GetChiSquared() - f.GetChiSquared()) < 1e-7 * f.GetChiSquared()
    assert a.GetPotentialOutlierCount() == f.GetPotentialOutlierCount()
    for suffix in ("rva", "pam"):
        for (ta, ra, ca, da), (tf, rf, cf, df) in zip(F.read_mtx(tmp_path / "multi" / f"n-{suffix}.mtx", 8), F.read_mtx(tmp_path / "single" / f"n-{suffix}.mtx", 8)):
            assert (ta, ra, ca) == (tf, rf, cf)
            assert np.abs(da - df).max() <= 1e-9 * np.abs(df).max()
    # again on the resident data (what bench.py times), then a cancelled run: every rank must leave the loop at the same point
    a.ResetAdjustment()
    assert a.AdjustNetworkDistributed() == st
    a.close()
    f.close()
    o.close()


def test_split_inverse_over_rccl(built, tmp_path):
    """simultaneous adjustment of one block on 2 GPUs: every large product of the blocked inverse split by tile columns, the parts
    exchanged by ncclBroadcast on the chain's stream (sym_inverse.hip gemm_split); against the one-GPU run of the same network"""
    _need(2)
    adjust.write_synthetic_network(str(tmp_path), "s", 50, 50, 0, 1, seed=5)       # n = 7 500: the larger launches of the recursion are split
    a = _run(str(tmp_path), "s", mode=adjust.SimultaneousMode, devices=[0, 1])
    st = a.AdjustNetworkDistributed()
    assert a.dist_info() == (0, 2, "rccl")
    ex = a.inverse_exchange_stats()
    assert ex["split_launches"] > 0 and ex["bytes_received"] > 0
    f = _run(str(tmp_path), "s", mode=adjust.SimultaneousMode)
    assert f.AdjustNetwork() == st == adjust.ADJUST_SUCCESS
    assert np.abs(a.block_estimates(0) - f.block_estimates(0)).max() < TOL_X
    va, vf = a.block_variances_packed(0), f.block_variances_packed(0)
    assert np.abs(va - vf).max() / np.abs(vf).max() < 1e-11
    a.close()
    f.close()


def test_process_per_gpu_bootstrap(built, orc, tmp_path):
    """two processes, one GPU each, launched the way mpirun / torchrun would (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT): the
    library's own TCP hand-off of the ncclUniqueId, ncclCommInitRank from two processes, the variance matrices to rank 0 by
    ncclSend / ncclRecv"""
    _need(2)
    adjust.write_synthetic_network(str(tmp_path), "n", 30, 12, 0, 6, seed=10)
    os.makedirs(tmp_path / "out", exist_ok=True)
    net = orc.Network(str(tmp_path / "n"), True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi_gpu_worker.py")
    port = 29000 + (os.getpid() % 1000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), "n"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the two ranks did not finish within 300 s")
        outs.append(out.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = np.load(tmp_path / "result.npz")
    assert int(res["status"]) == ost and int(res["iterations"]) == o.iterations() and int(res["rccl_ranks"]) == 2
    assert sorted(set(res["owners"].tolist())) == [0, 1]
    for k in range(6):
        assert np.abs(res[f"est_{k}"] - o.block_estimates(k)).max() < TOL_X
    for (t, r, c, d), k in zip(F.read_mtx(tmp_path / "out" / "n-rva.mtx", 6), range(6)):
        vo = o.block_variances(k)
        assert np.abs(d - vo).max() / np.abs(vo).max() < TOL_V          # blocks of rank 1 included: they travelled to rank 0
    o.close()
