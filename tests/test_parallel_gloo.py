"""The N > 1 SCHEDULE on CPU: tests/parallel_harness.py (round 1's Python orchestrator over the per-block C entry points: ownership,
junction / condensed-block exchange, coordinate all_reduce) driving tests/numpy_backend.py (a numpy restatement of the block steps)
under gloo with world sizes 2 and 3, against the CPU oracle's single-process phased adjustment.

What this pins: the exchange pattern and the algebra of the one- and two-level condensed chains across ranks.  What it does NOT run: the
product's C++ driver (dna_adjust::AdjustPhasedDistributed, host/dna_adjust_dist.cpp) -- that needs a device; its schedule (block owners,
run boundaries, message sizes) is pinned device-free by tests/test_dist_plan.py through dnaadj_dist_plan, and its arithmetic on the
GPU by tests/test_gpu_distributed.py (ranks as threads) and tests/test_gpu_multi.py (RCCL, one rank per GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, base, outdir, condensed):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import parallel_harness as parallel
    from tests import oracle
    from tests.numpy_backend import NumpyBlockBackend
    net = oracle.Network(base, True)
    be = NumpyBlockBackend(net, condensed=condensed)
    status, its, corr = parallel.run_phased(be, dist, rank, world)
    if condensed:
        owner = parallel.block_owners([float(be.n_stations(k)) ** 3 for k in range(be.n_blocks)], world)
        final_owner = lambda k: owner[k]
    else:
        final_owner = parallel.PhasedSchedule([be.flags(k) for k in range(be.n_blocks)], world).final_owner
    res = {"status": status, "iterations": its, "corrections": np.array(corr)}
    for k in range(be.n_blocks):
        res[f"coords_{k}"] = be.blk[k]["rig"]          # every rank must hold the rigorous coordinates of every block
        if final_owner(k) == rank:
            res[f"owned_var_{k}"] = be.blk[k]["rigvar"]
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,blocks,condensed", [(2, 5, True), (3, 6, True), (2, 2, True), (2, 5, False), (3, 6, False), (2, 2, False)])
def test_distributed_schedule_matches_single_process(built, orc, tmp_path, world, blocks, condensed):
    """both multi-rank schedules -- the condensed one (every block reduced to its shared stations, chains on the reduced
    blocks, one rigorous solve per block; the default) and the reference's (forward chain || reverse chain, combination
    solves spread) -- against the oracle's single-process phased adjustment"""
    from dynadjust_amd import adjust
    from dynadjust_amd.device import unpack_lower
    adjust.write_synthetic_network(str(tmp_path), "n", 12, 6, 0, blocks, seed=3 + blocks)
    base = str(tmp_path / "n")
    net = orc.Network(base, True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    mp.spawn(_worker, args=(world, _free_port(), base, str(tmp_path), condensed), nprocs=world, join=True)
    owned = set()
    for r in range(world):
        res = np.load(str(tmp_path / f"rank{r}.npz"))
        assert int(res["status"]) == ost and int(res["iterations"]) == o.iterations()
        for i in range(o.iterations()):
            assert abs(res["corrections"][i] - o.max_correction(i + 1)) < 1e-8
        for k in range(blocks):
            assert np.abs(res[f"coords_{k}"] - o.block_estimates(k)).max() < 1e-8
            if f"owned_var_{k}" in res:
                owned.add(k)
                n = 3 * len(o.block_stations(k))
                V = unpack_lower(o.block_variances(k), n)
                assert np.abs(res[f"owned_var_{k}"] - V).max() / np.abs(V).max() < 1e-8
    assert owned == set(range(blocks))      # every block's rigorous variances live on exactly one rank... or more
    o.close()


def _two_level_worker(rank, world, port, base, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle
    from tests.numpy_backend import NumpyBlockBackend, run_two_level
    be = NumpyBlockBackend(oracle.Network(base, True))
    status, its, corr, owner = run_two_level(be, dist, rank, world)
    res = {"status": status, "iterations": its, "corrections": np.array(corr), "owner": np.array(owner)}
    for k in range(be.n_blocks):
        res[f"coords_{k}"] = be.blk[k]["rig"]
        if owner[k] == rank:
            res[f"owned_var_{k}"] = be.blk[k]["rigvar"]
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,blocks", [(2, 5), (3, 7), (3, 3)])
def test_two_level_chains_match_single_process(built, orc, tmp_path, world, blocks):
    """the two-level chains of the C++ multi-GPU driver (dna_adjust_dist.cpp: every rank reduces its own run of condensed blocks to
    the stations of the run's ends, ONE system per rank is broadcast, the chains run over the runs and then inside every run), restated
    densely in numpy and run under gloo: same coordinates and variances as the oracle's single-process phased adjustment"""
    from dynadjust_amd import adjust
    from dynadjust_amd.device import unpack_lower
    adjust.write_synthetic_network(str(tmp_path), "n", 3 * blocks, 6, 0, blocks, seed=11 + blocks)
    base = str(tmp_path / "n")
    net = orc.Network(base, True)
    o = orc.Adjustment(net, True)
    o.prepare()
    ost = o.run()
    mp.spawn(_two_level_worker, args=(world, _free_port(), base, str(tmp_path)), nprocs=world, join=True)
    owned = set()
    for r in range(world):
        res = np.load(str(tmp_path / f"rank{r}.npz"))
        owner = list(res["owner"])
        assert owner == sorted(owner) and set(owner) == set(range(world))          # contiguous runs, every rank one
        assert int(res["status"]) == ost and int(res["iterations"]) == o.iterations()
        for i in range(o.iterations()):
            assert abs(res["corrections"][i] - o.max_correction(i + 1)) < 1e-8
        for k in range(blocks):
            assert np.abs(res[f"coords_{k}"] - o.block_estimates(k)).max() < 1e-8
            if f"owned_var_{k}" in res:
                owned.add(k)
                V = unpack_lower(o.block_variances(k), 3 * len(o.block_stations(k)))
                assert np.abs(res[f"owned_var_{k}"] - V).max() / np.abs(V).max() < 1e-8
    assert owned == set(range(blocks))
    o.close()


def test_contiguous_owners():
    from tests.numpy_backend import contiguous_owners
    assert contiguous_owners([1.0] * 16, 8) == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7]
    assert contiguous_owners([1.0] * 5, 1) == [0] * 5
    assert contiguous_owners([8.0, 1, 1, 1, 1, 1, 1, 1, 1], 2) == [0] + [1] * 8
    o = contiguous_owners([1.0] * 6, 8)
    assert o == [0, 1, 2, 3, 4, 5]                                   # more ranks than blocks: one block each
    o = contiguous_owners([0.5, 1, 1, 1, 1, 1, 1, 0.5], 3)
    assert o == sorted(o) and set(o) == {0, 1, 2}


def test_schedule_roles():
    from tests.parallel_harness import PhasedSchedule
    flags = [(True, False, False)] + [(False, False, False)] * 6 + [(False, True, False)]
    s = PhasedSchedule(flags, 4)
    assert s.intermediate == [1, 2, 3, 4, 5, 6]
    assert [s.combine_owner[k] for k in s.intermediate] == [0, 1, 2, 3, 0, 1]      # every rank takes combination solves
    assert s.final_owner(0) == s.rev_rank == 1 and s.final_owner(7) == s.fwd_rank == 0
    s1 = PhasedSchedule(flags, 1)
    assert s1.rev_rank == 0 and set(s1.combine_owner.values()) == {0}
    iso = PhasedSchedule([(True, True, True), (True, False, False), (False, True, False)], 2)
    assert iso.intermediate == [] and iso.final_owner(0) == 0 and iso.final_owner(1) == 1 and iso.final_owner(2) == 0


def test_block_owners_balance():
    from tests.parallel_harness import block_owners
    assert block_owners([1.0] * 16, 8) == [0, 1, 2, 3, 4, 5, 6, 7] * 2
    assert block_owners([1.0] * 5, 1) == [0] * 5
    o = block_owners([8.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0], 2)
    assert o[0] == 0 and o.count(1) == 8                       # the big block alone, the small ones together
    assert block_owners([3.0, 2.0, 2.0], 4) == [0, 1, 2]
