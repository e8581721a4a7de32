#!/usr/bin/env python
"""The serial remainder of the condensed schedule at BASELINE.json configs[3] size (1 000 x 1 000 stations, 128 blocks, junction rows
of 1 000 stations = 3 000 unknowns, condensed blocks of 6 000), measured on ONE GPU.

cfg4 itself needs 4+ GPUs for its rigorous variances, but its chain phase only depends on the junction rows: a network of 128
strips of TWO grid rows x 1 000 columns has the same 127 junction rows of 1 000 stations (condensed blocks of 6 000 unknowns) and
cheap large phases.  Measured:

  one-level   every rank runs both chains over all 128 condensed blocks (2 x 127 steps, forward || reverse on two chains) and every
              condensed block is broadcast to every rank: chain time of one rank, payload bytes per rank;
  two-level   (a.dist_two_level) 8 ranks own runs of 16 blocks: level 1 (own run -> end stations), exchange (one system per rank),
              level 2 (chains over 8 runs), level 3 (chains over the own 16 blocks): per-rank chain time with the ranks taking the one GPU
              in turns (DNAGPU_LOCAL_EXCLUSIVE=1), payload bytes per rank.

Writes one JSON record (gpurun_out/chain_phase_cfg4.json).  The large phases of a cfg4 iteration on 8 GPUs -- 16 blocks of n ~ 27 000 per
GPU -- are bench.py --workload cfg4_slice's step time / iterations (profiles/): the share of the chain phase follows."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["DNAGPU_LOCAL_EXCLUSIVE"] = "1"
os.environ.setdefault("DNAGPU_CHAINS", "2")        # (8 ranks share one GPU here: two chains each keep the workspaces at 33 GB)

from dynadjust_amd import adjust  # noqa: E402


def run(folder, ranks, two_level, steps=2):
    # (no kept factors: eight instances budgeting the same GPU's free memory independently would over-commit it)
    kw = dict(adjust_mode=adjust.PhasedMode, multi_thread=True, max_iterations=1, keep_factors=False)
    if ranks > 1:
        kw.update(devices=[0] * ranks, dist_transport="local", dist_two_level=two_level)
    else:
        os.environ["DNAGPU_FORCE_DISTRIBUTED"] = "1"
    a = adjust.DnaAdjust()
    a.PrepareAdjustment(adjust.ProjectSettings("net", folder, **kw))
    rec = []
    for _ in range(steps):
        a.ResetAdjustment()
        b0 = a.exchange_stats()["bytes"]
        t0 = time.perf_counter()
        a.AdjustNetwork()
        dt = time.perf_counter() - t0
        ex = a.exchange_stats()
        rec.append({"iteration_s": dt, "chain_ms": ex["chain_ms"], "exchange_ms": ex["exchange_ms"], "payload_bytes": ex["bytes"] - b0})
    a.close()
    os.environ.pop("DNAGPU_FORCE_DISTRIBUTED", None)
    return rec[-1]


def main():
    blocks = int(os.environ.get("CHAIN_BLOCKS", "128"))
    cols = int(os.environ.get("CHAIN_COLS", "1000"))
    d = tempfile.mkdtemp(prefix="dnagpu_chain_")
    info = adjust.write_synthetic_network(d, "net", 2 * blocks, cols, 0, blocks)
    one = run(d, 1, False)
    two = run(d, 8, True)
    xgmi = 50e9     # assumed effective bytes / s into one GPU during a broadcast over xGMI (7 links x 153 GB/s nominal; stated, not measured)
    out = {
        "network": {"stations": info["stations"], "blocks": blocks, "junction_row_stations": cols, "condensed_block_unknowns": 6 * cols},
        "one_level": {"chain_phase_ms_per_rank": one["chain_ms"], "payload_bytes_received_per_rank": one["payload_bytes"],
                      "exchange_model_ms_at_50GBps": one["payload_bytes"] * (7.0 / 8.0) / xgmi * 1e3,
                      "note": "payload counted on the one rank of this run = all 128 condensed blocks; on 8 ranks each receives 7/8 of it"},
        "two_level": {"chain_phase_ms_rank0": two["chain_ms"], "payload_bytes_rank0": two["payload_bytes"],
                      "exchange_model_ms_at_50GBps": two["payload_bytes"] / xgmi * 1e3,
                      "note": "rank 0's levels 1-3 timed with the GPU to itself (ranks take turns); its own run system does not travel to it"},
    }
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "chain_phase_cfg4.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
