#!/usr/bin/env python
"""Look-ahead inside the factorisation of one matrix (dnagpu_ctx_set_lookahead) on 1 x MI355X, through the C-ABI, HIP resident:
inverse / eliminate / keep + finish of tools/gpu_inverse_bench.py with look-ahead off and on (side launches holding the given shares
of the chip's workgroup slots), best of 3 timed repetitions each, and the inverse compared BIT FOR BIT between the two.
usage: python tools/gpu_lookahead_probe.py [--shares 50,75,88] [stations ...]   (n = 3 x stations)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd.device import DeviceContext
from tools.gpu_inverse_bench import spd_packed


def main():
    args = sys.argv[1:]
    shares = [75]
    if args and args[0] == "--shares":
        shares = [int(x) for x in args[1].split(",")]
        args = args[2:]
    rng = np.random.default_rng(1)
    print("# n, mode: inverse ms / TFLOP/s (n^3) | eliminate ms / TFLOP/s (n^3/3) | keep+finish ms / TFLOP/s (n^3) | nodes that looked ahead | inverse equal bit for bit")
    with DeviceContext(0) as ctx:
        lib = ctx.lib
        for ns in ([int(a) for a in args] or [2048, 4096, 6656, 10000]):
            n = 3 * ns
            ap = spd_packed(n, rng)
            m = ctx.matrix(n)
            ctx.block_create(0, ns, 0)
            ctx.block_set_stations(0, np.zeros(3 * ns))
            keep = np.arange(ns - max(1, ns // 100), ns, dtype=np.uint32)
            red = ctx.matrix(3 * len(keep))
            inv = ctx.matrix(n)
            pf = ctx.partial_create(n, 3 * len(keep))
            ref = None
            for mode in [None] + shares:
                lib.dnagpu_ctx_set_lookahead(ctx.h, 0 if mode is None else 1)
                if mode is not None:
                    lib.dnagpu_debug_set_side_share(mode)
                nodes0 = lib.dnagpu_lookahead_nodes(ctx.h)
                res = []
                same = ""
                for what in ("inverse", "eliminate", "keep"):
                    best = 1e9
                    for rep in range(4):
                        m.upload_packed(ap, n)
                        ctx.sync()
                        t0 = time.perf_counter()
                        if what == "inverse":
                            m.invert()
                        elif what == "eliminate":
                            ctx.block_reduce(0, m, keep, red)
                        else:
                            ctx.block_reduce(0, m, keep, red, keep=pf)
                            ctx.partial_complete(pf, red, inv, n)
                        ctx.sync()
                        if rep:
                            best = min(best, time.perf_counter() - t0)
                    if what == "inverse":
                        got = m.download_packed()
                        if mode is None:
                            ref = got
                        else:
                            same = "equal" if np.array_equal(got, ref) else "DIFFERENT (max |d| %.3e)" % float(np.max(np.abs(got - ref)))
                    flops = float(n) ** 3 / (3.0 if what == "eliminate" else 1.0)
                    res.append("%8.2f ms %6.1f" % (best * 1e3, flops / best / 1e12))
                print("n = %6d %-9s: %s | %s | %s | %5d | %s" % (n, "off" if mode is None else "share %d" % mode, *res,
                                                               lib.dnagpu_lookahead_nodes(ctx.h) - nodes0, same), flush=True)
            lib.dnagpu_ctx_set_lookahead(ctx.h, 0)
            lib.dnagpu_debug_set_side_share(0)
            ctx.partial_destroy(pf)
            for q in (m, red, inv):
                q.close()
            ctx.block_destroy(0)


if __name__ == "__main__":
    main()
