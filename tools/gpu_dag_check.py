#!/usr/bin/env python
"""The tile-DAG path against the per-product path on the device: same inputs through dnagpu_invert / dnagpu_block_reduce with
DNAGPU_DAG on and off, results compared element by element, every repetition timed, the ticket counter checked."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd.device import DeviceContext
from tools.gpu_inverse_bench import spd_packed


def state(ctx):
    l, t = C.c_uint64(), C.c_uint64()
    ctx.lib.dnagpu_tile_dag_stats(ctx.h, C.byref(l), C.byref(t))
    return l.value, t.value


def main():
    rng = np.random.default_rng(1)
    sizes = [int(s) for s in sys.argv[1:]] or [1024, 2048, 4096]
    with DeviceContext(0) as ctx:
        for ns in sizes:
            n = 3 * ns
            ap = spd_packed(n, rng)
            m = ctx.matrix(n)
            ctx.block_create(0, ns, 0)
            ctx.block_set_stations(0, np.zeros(3 * ns))
            keep = np.arange(ns - max(1, ns // 100), ns, dtype=np.uint32)
            red = ctx.matrix(3 * len(keep))
            ref = {}
            for dag in (0, 1):
                ctx.lib.dnagpu_debug_set_tile_dag(dag)
                for what in (os.environ.get("DAG_CHECK_WHAT", "inverse,eliminate").split(",")):
                    times = []
                    for rep in range(int(os.environ.get("DAG_CHECK_REPS", "3"))):
                        m.upload_packed(ap, n)
                        ctx.sync()
                        t0 = time.perf_counter()
                        try:
                            if what == "inverse":
                                m.invert()
                                out = None
                            else:
                                ctx.block_reduce(0, m, keep, red)
                            ctx.sync()
                            err = ""
                        except Exception as e:
                            err = " ERROR " + str(e)
                        times.append((time.perf_counter() - t0) * 1e3)
                        out = m.download_packed() if what == "inverse" else red.download_packed()
                        if dag == 0:
                            ref[what] = out
                        else:
                            d = float(np.abs(out - ref[what]).max() / np.abs(ref[what]).max())
                            print(f"   n {n} {what} rep {rep}: {times[-1]:.2f} ms, rel diff vs per-product path {d:.2e}, DAG (launches, tasks) {state(ctx)}{err}", flush=True)
                    if dag == 0:
                        print(f"n {n} {what} per-product path: {min(times):.2f} ms", flush=True)
            m.close(); red.close()
            ctx.block_destroy(0)


if __name__ == "__main__":
    main()
