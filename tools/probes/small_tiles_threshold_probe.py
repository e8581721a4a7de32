import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from dynadjust_amd.device import DeviceContext
from tools.gpu_inverse_bench import spd_packed
rng = np.random.default_rng(1)
with DeviceContext(0) as ctx:
    lib = ctx.lib
    for ns in (2048, 4096, 6656, 10000):
        n = 3 * ns
        ap = spd_packed(n, rng)
        m = ctx.matrix(n)
        ctx.block_create(0, ns, 0)
        ctx.block_set_stations(0, np.zeros(3 * ns))
        keep = np.arange(ns - max(1, ns // 100), ns, dtype=np.uint32)
        red = ctx.matrix(3 * len(keep))
        for thr in (512, 768, 1024, 1536, 2304, 4096):
            old = lib.dnagpu_debug_set_small_tiles(thr)
            res = []
            for what in ("inverse", "eliminate"):
                best = 1e9
                for rep in range(4):
                    m.upload_packed(ap, n)
                    ctx.sync()
                    t0 = time.perf_counter()
                    if what == "inverse":
                        m.invert()
                    else:
                        ctx.block_reduce(0, m, keep, red)
                    ctx.sync()
                    if rep:
                        best = min(best, time.perf_counter() - t0)
                flops = float(n) ** 3 / (3.0 if what == "eliminate" else 1.0)
                res.append("%8.2f ms %6.1f" % (best * 1e3, flops / best / 1e12))
            print("n = %6d  small-launch threshold %5d tiles: inverse %s | eliminate %s" % (n, thr, *res), flush=True)
            lib.dnagpu_debug_set_small_tiles(-1)
        for q in (m, red):
            q.close()
        ctx.block_destroy(0)
