#!/usr/bin/env python
"""Wall time of the three phases of one condensed iteration on cfg3 (1 x MI355X): (A) condense, (B) chains on the condensed
blocks, (C) rigorous solves -- through the same C entry points the multi-GPU orchestrator uses."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd import adjust
from tests import parallel_harness as parallel
import torch

d = tempfile.mkdtemp()
adjust.write_synthetic_network(d, "net", 316, 317, 266666, 16)
p = adjust.ProjectSettings("net", d, adjust_mode=adjust.PhasedMode, multi_thread=bool(int(os.environ.get("DNAGPU_MULTI_THREAD", "1"))))
be = parallel.DeviceBlockBackend(p, torch.device("cpu"))
blocks = list(range(be.n_blocks))
for it in range(3):
    be.adj.ResetAdjustment()
    be.begin_iteration()
    t0 = time.perf_counter(); be.condense_blocks(blocks)
    t1 = time.perf_counter(); be.condensed_chains()
    t2 = time.perf_counter(); be.rigorous_blocks(blocks)
    t3 = time.perf_counter()
    print("iteration %d: condense %.3f s, chains %.3f s, rigorous %.3f s" % (it, t1 - t0, t2 - t1, t3 - t2), flush=True)
be.close()
