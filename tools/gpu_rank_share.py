#!/usr/bin/env python
"""MODEL of the N-GPU step, measured on ONE GPU: what rank 0 of an N-GPU cfg3 run does -- its share of the 16 blocks condensed (A) and solved
(C) in iteration 1, the same with the kept factors in iteration 2 (a.reuse_factors: right-hand sides only), the chains on ALL condensed
blocks (B, one level), the variance matrices of its blocks after the last iteration (D).  The exchange between ranks is excluded
(cfg3: 16 condensed blocks of 29 MB per iteration).  No run with more than one GPU has been possible on this pool: the figures below are
what such a run cannot beat, not a measurement of it.
    python tools/gpu_rank_share.py [N ...]"""
import os, sys, time, tempfile
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd import adjust
from tests import parallel_harness as parallel
import torch

d = tempfile.mkdtemp()
adjust.write_synthetic_network(d, "net", 316, 317, 266666, 16)
p = adjust.ProjectSettings("net", d, adjust_mode=adjust.PhasedMode, multi_thread=True)
be = parallel.DeviceBlockBackend(p, torch.device("cpu"))
B = be.n_blocks
allb = list(range(B))
base = None
print("# python tools/gpu_rank_share.py on 1 x MI355X (round 5): MODEL, not a multi-GPU measurement -- rank 0's work of an N-GPU cfg3 step, on one GPU")
for world in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    owner = parallel.block_owners([float(be.n_stations(k)) ** 3 for k in range(B)], world)
    mine = [k for k in allb if owner[k] == 0]
    others = [k for k in allb if k not in mine]
    best = None
    for rep in range(3):
        be.adj.ResetAdjustment()
        t = []
        for it in range(2):
            be.begin_iteration()
            be.condense_blocks(others)      # the other ranks' blocks (untimed: their condensed blocks arrive by broadcast)
            t0 = time.perf_counter(); be.condense_blocks(mine)
            t1 = time.perf_counter(); be.condensed_chains()
            t2 = time.perf_counter(); be.rigorous_blocks(mine)
            t3 = time.perf_counter()
            t.append((t1 - t0, t2 - t1, t3 - t2))
            if it == 0:
                be.end_iteration()
        tv = time.perf_counter()
        be.finish()                         # the variance matrices of the blocks solved above
        tv = time.perf_counter() - tv
        step = sum(sum(x) for x in t) + tv
        if rep and (best is None or step < best[0]):
            best = (step, t, tv)
    step, t, tv = best
    base = base or step
    print("N = %d: rank 0 owns %2d blocks: iteration 1 condense %.3f + chains %.3f + solve %.3f s; iteration 2 (kept factors) %.3f + %.3f + %.3f s; "
          "variance matrices %.3f s; step %.3f s -> %.2fx" % (world, len(mine), t[0][0], t[0][1], t[0][2], t[1][0], t[1][1], t[1][2], tv, step, base / step), flush=True)
be.close()
