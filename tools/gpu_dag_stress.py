#!/usr/bin/env python
"""Repeats DAG launches and counts the ones whose result differs from the per-product path's (a race shows as a count > 0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd.device import DeviceContext
from tools.gpu_inverse_bench import spd_packed

reps = int(os.environ.get("STRESS_REPS", "30"))
rng = np.random.default_rng(1)
with DeviceContext(0) as ctx:
    for ns in [int(s) for s in sys.argv[1:]] or [2048, 4096]:
        n = 3 * ns
        ap = spd_packed(n, rng)
        m = ctx.matrix(n)
        ctx.block_create(0, ns, 0)
        ctx.block_set_stations(0, np.zeros(3 * ns))
        keep = np.arange(ns - max(1, ns // 100), ns, dtype=np.uint32)
        red = ctx.matrix(3 * len(keep))
        ref = {}
        ctx.lib.dnagpu_debug_set_tile_dag(0)
        m.upload_packed(ap, n); m.invert(); ref["inverse"] = m.download_packed()
        m.upload_packed(ap, n); ctx.block_reduce(0, m, keep, red); ref["eliminate"] = red.download_packed()
        ctx.lib.dnagpu_debug_set_tile_dag(1)
        for what in ("inverse", "eliminate", "inverse"):
            bad, errs, t = 0, 0, 0.0
            for rep in range(reps):
                m.upload_packed(ap, n)
                ctx.sync()
                t0 = time.perf_counter()
                try:
                    if what == "inverse":
                        m.invert()
                    else:
                        ctx.block_reduce(0, m, keep, red)
                    ctx.sync()
                except Exception as e:
                    errs += 1
                    continue
                t += time.perf_counter() - t0
                out = m.download_packed() if what == "inverse" else red.download_packed()
                if not np.array_equal(out, ref[what]):
                    bad += 1
            print(f"n {n} {what}: {reps} launches, {bad} differ, {errs} errors, {t / max(1, reps - errs) * 1e3:.2f} ms each", flush=True)
        m.close(); red.close(); ctx.block_destroy(0)
