#!/usr/bin/env python
"""What one rank of an N-GPU run does per iteration, measured on one GPU: cfg3's 16 blocks, the rank's share condensed and
solved (LPT owners of parallel.block_owners), the chains on all condensed blocks -- the exchange between ranks excluded.
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
for world in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    owner = parallel.block_owners([float(be.n_stations(k)) ** 3 for k in range(B)], world)
    mine = [k for k in allb if owner[k] == 0]
    best = None
    for rep in range(3):
        be.adj.ResetAdjustment()
        be.begin_iteration()
        be.condense_blocks([k for k in allb if k not in mine])      # the other ranks' blocks (untimed: their payloads arrive by broadcast)
        t0 = time.perf_counter(); be.condense_blocks(mine)
        t1 = time.perf_counter(); be.condensed_chains()
        t2 = time.perf_counter(); be.rigorous_blocks(mine)
        t3 = time.perf_counter()
        be.finish()                                                   # the variance matrices of the blocks solved above (a.defer_variances)
        t4 = time.perf_counter()
        cur = (t3 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3)
        if rep and (best is None or cur[0] + cur[4] < best[0] + best[4]):
            best = cur
    print("N = %d: rank 0 owns %d blocks: condense %.3f s, chains %.3f s, solve %.3f s -> %.3f s per iteration; variance matrices at the end %.3f s; "
          "a step of 2 iterations: %.3f s" % (world, len(mine), best[1], best[2], best[3], best[0], best[4], 2 * best[0] + best[4]), flush=True)
be.close()
