"""The N > 1 schedule of the product's C++ driver (host/dna_adjust_dist.cpp) pinned WITHOUT a device: dnaadj_plan_distributed runs
PrepareAdjustment's host side and memory plan for every rank of a world of GPUs it does not have -- ComputeBlockOwners, DecideStaging,
PrepareCondensedBlocks, PrepareTwoLevel are the functions the adjustment itself runs -- and returns what they decided.  (The reference sizes
its threads' work before it starts them: dnaadjust-multi.cpp:92-140; its blocks never leave one address space.)"""
import itertools

import numpy as np
import pytest

from dynadjust_amd import adjust
from tests import dnaformats as F


def _plan(folder, world, hbm=309.0e9, **kw):
    a = adjust.DnaAdjust()
    p = adjust.ProjectSettings("n", folder, adjust_mode=adjust.PhasedMode, multi_thread=True, **kw)
    try:
        return a.plan_distributed(p, world, hbm)
    finally:
        a.close()


def _best_contiguous(costs, world):
    """the smallest possible largest run over all cuts of the block sequence into `world` non-empty runs (brute force)"""
    B = len(costs)
    best = None
    for cuts in itertools.combinations(range(1, B), world - 1):
        edges = (0,) + cuts + (B,)
        worst = max(sum(costs[edges[i]:edges[i + 1]]) for i in range(world))
        best = worst if best is None else min(best, worst)
    return best


@pytest.mark.parametrize("rows,cols,blocks,world,kw", [(48, 20, 8, 4, {}), (45, 12, 9, 2, {}), (60, 10, 10, 3, {"ragged": 0.5}), (40, 8, 5, 5, {})])
def test_block_owners_are_contiguous_balanced_runs(built, tmp_path, rows, cols, blocks, world, kw):
    """ComputeBlockOwners (condensed schedule): every rank a contiguous run of blocks, the largest run's sum of n^3 as small as any cut can make
    it; identical on every rank"""
    info = adjust.write_synthetic_network(str(tmp_path), "n", rows, cols, 0, blocks, seed=5, **kw)
    plan = _plan(str(tmp_path), world)
    assert plan["world"] == world and plan["blocks"] == info["blocks"] and len(plan["ranks"]) == world
    owners = plan["ranks"][0]["owners"]
    for r in plan["ranks"]:
        assert r["owners"] == owners and r["condensed_schedule"]
        assert [k for k, o in enumerate(owners) if o == r["rank"]] == list(range(r["first_block"], r["last_block"] + 1))
        assert r["own_blocks"] == r["last_block"] - r["first_block"] + 1 >= 1
    assert owners == sorted(owners) and set(owners) == set(range(world))
    ISL, JSL, _, _ = F.read_seg(str(tmp_path / "n.seg"))
    shares = [r["share_of_sum_n3"] for r in plan["ranks"]]
    assert abs(sum(shares) - 1.0) < 1e-9
    costs = [float(3 * (len(i) + len(j))) ** 3 for i, j in zip(ISL, JSL)]
    assert abs(max(shares) * sum(costs) - _best_contiguous(costs, world)) < 1e-6 * sum(costs)


def test_two_level_runs_and_exchange_sizes(built, tmp_path):
    """PrepareTwoLevel: every rank's run condensed to the junction rows at its two ends, merged block by block in order; level 2 = world - 1
    steps each way, level 3 = the own blocks; the bytes of an iteration's exchanges follow from the padded orders"""
    cols, world = 20, 4
    adjust.write_synthetic_network(str(tmp_path), "n", 48, cols, 0, 8, seed=3)
    plan = _plan(str(tmp_path), world)
    pad = lambda n: -(-n // 128) * 128
    run_bytes = 0
    for r in plan["ranks"]:
        assert r["two_level_chains"]
        run = r["run"]
        assert (run["a"], run["b"]) == (r["first_block"], r["last_block"])
        inner = 0 < r["rank"] < world - 1
        assert run["towards_previous"] == (cols if r["rank"] > 0 else 0) and run["towards_next"] == (cols if r["rank"] < world - 1 else 0)
        assert run["end_stations"] == (2 * cols if inner else cols)
        assert [m["block"] for m in run["merges"]] == list(range(run["a"] + 1, run["b"] + 1))
        assert run["merges"][-1]["stay"] == run["end_stations"]
        assert run["level2_steps_each_way"] == world - 1 and run["level3_steps_each_way"] == run["b"] - run["a"]
        run_bytes += (pad(3 * run["end_stations"]) ** 2 + pad(3 * run["end_stations"])) * 8
    for r in plan["ranks"]:
        ex = r["exchange_bytes_per_iteration"]
        assert ex["run_systems_two_level"] == run_bytes
        # one level: every condensed block (its kept stations: one junction row for the end blocks, two for the others) to every rank
        kept = [cols] + [2 * cols] * 6 + [cols]
        assert ex["condensed_blocks_one_level"] == sum((pad(3 * k) ** 2 + pad(3 * k)) * 8 for k in kept)
        assert ex["coordinates_all_reduce"] == 8 * (world + 3 * (48 * cols + 7 * cols))      # every block's stations, junction rows twice
    one = _plan(str(tmp_path), world, dist_two_level=False)
    assert not any(r["two_level_chains"] for r in one["ranks"])


def test_the_memory_plan_follows_the_gpu_it_is_given(built, tmp_path):
    """the same network on GPUs of different size: everything resident; variance matrices staged (host memory first) once they do not fit;
    kept factors and batch members only as far as the budget goes -- what DecideStaging / PrepareCondensedBlocks decide"""
    adjust.write_synthetic_network(str(tmp_path), "n", 120, 40, 0, 6, seed=3)      # blocks of n ~ 2 500
    big = _plan(str(tmp_path), 2, 309.0e9)["ranks"][0]
    assert big["hbm"]["fits"] and not big["staged"] and big["blocks_keeping_their_factor"] == big["own_blocks"] and not big["factors_made_again"]
    need = big["hbm"]["committed"]
    small = _plan(str(tmp_path), 2, 1.5e9 + 0.6 * need)["ranks"][0]
    assert small["staged"] and small["hbm"]["staged_in_host_memory"] + small["hbm"]["variance_matrices"] > 0
    assert small["hbm"]["committed"] < need
    tiny = _plan(str(tmp_path), 2, 1.5e9 + 0.3 * need)["ranks"][0]
    assert tiny["staged"] and tiny["blocks_keeping_their_factor"] < tiny["own_blocks"]
    assert tiny["factors_made_again"]            # (GNSS only: the blocks without a kept factor make it again instead of inverting every iteration)
    # a plan leaves the handle unprepared and usable
    a = adjust.DnaAdjust()
    p = adjust.ProjectSettings("n", str(tmp_path), adjust_mode=adjust.PhasedMode)
    a.plan_distributed(p, 3)
    with pytest.raises(adjust.NetAdjustException):
        a.AdjustNetwork()
    a.close()
