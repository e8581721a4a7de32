#!/usr/bin/env python
"""Intra-block distributed inverse (dnagpu_set_inverse_exchange) on BASELINE.json configs[1] (cfg2: 10 000 stations, simultaneous, one block of
n = 29 988): the exchange volume per rank and the correctness of the split, measured with W ranks as threads SHARING one GPU (transport
"local": every part crosses the device once per receiving rank, so the time on one GPU says nothing about W GPUs; the bytes do).

Prints / writes gpurun_out/intra_block_cfg2.json: per W the launches that were split, the bytes one rank receives per inverse, the
single-GPU step time for reference, and the largest deviation from the single-GPU result (must be 0: same tile code on the same operands)."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402
from dynadjust_amd import adjust  # noqa: E402


def main():
    d = tempfile.mkdtemp(prefix="dnagpu_intra_")
    info = adjust.write_synthetic_network(d, "net", 100, 100, 26666, 1)
    out = {"workload": "cfg2", "unknowns": int(info["max_block_unknowns"]), "runs": []}
    ref = None
    for W in (1, 2, 4):
        kw = dict(adjust_mode=adjust.SimultaneousMode)
        if W > 1:
            kw.update(devices=[0] * W, dist_transport="local")
        a = adjust.DnaAdjust()
        a.PrepareAdjustment(adjust.ProjectSettings("net", d, **kw))
        a.AdjustNetwork()                       # warm-up (tables, workspaces)
        a.ResetAdjustment()
        ex0 = a.inverse_exchange_stats() if W > 1 else {"split_launches": 0, "bytes_received": 0.0}
        t0 = time.perf_counter()
        st = a.AdjustNetwork()
        dt = time.perf_counter() - t0
        ex1 = a.inverse_exchange_stats() if W > 1 else ex0
        x = a.block_estimates(0)
        v = a.block_variances_packed(0)
        if ref is None:
            ref = (x, v)
        out["runs"].append({"ranks": W, "status": st, "step_s_on_one_shared_gpu": dt,
                            "split_launches_per_inverse": ex1["split_launches"] - ex0["split_launches"],
                            "bytes_received_per_rank_per_inverse": ex1["bytes_received"] - ex0["bytes_received"],
                            "max_abs_dx_vs_one_gpu": float(np.abs(x - ref[0]).max()), "max_abs_dvar_vs_one_gpu": float(np.abs(v - ref[1]).max())})
        a.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "intra_block_cfg2.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
