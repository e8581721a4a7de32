#!/usr/bin/env python
"""Rates of the dense building blocks on 1 x MI355X, through the C-ABI (dnagpu.h), HIP resident, 3 repetitions each:
  inverse      dnagpu_invert: potrf + trtri + lauum, n^3 flops (the reference's Solve())
  eliminate    dnagpu_block_reduce onto 1 % of the unknowns: a Cholesky factorisation that stops before them, ~0.34 n^3 issued,
               priced at the factorisation's own n^3 / 3 ("potrf-only rate")
  keep+finish  dnagpu_block_reduce(keep) + dnagpu_partial_complete: the condensed schedule's pair, n^3 in all"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynadjust_amd.device import DeviceContext


def spd_packed(n, rng):
    # diagonally dominant SPD without forming n^2 on the host more than once
    d = 4.0 + rng.random(n)
    ap = np.zeros(n * (n + 1) // 2)
    off = 0
    for j in range(n):
        col = ap[off:off + n - j]
        col[0] = d[j] * 8.0
        m = min(n - j - 1, 6)
        if m:
            col[1:1 + m] = 0.5 * rng.random(m)
        off += n - j
    return ap


def main():
    rng = np.random.default_rng(1)
    print("# n, inverse ms / TFLOP/s (n^3), eliminate ms / TFLOP/s (n^3/3), keep+finish ms / TFLOP/s (n^3)")
    with DeviceContext(0) as ctx:
        for ns in ([int(a) for a in sys.argv[1:]] or [2048, 4096, 6656, 10000]):
            n = 3 * ns
            ap = spd_packed(n, rng)
            m = ctx.matrix(n)
            ctx.block_create(0, ns, 0)
            ctx.block_set_stations(0, np.zeros(3 * ns))
            keep = np.arange(ns - max(1, ns // 100), ns, dtype=np.uint32)
            red = ctx.matrix(3 * len(keep))
            inv = ctx.matrix(n)
            pf = ctx.partial_create(n, 3 * len(keep))
            res = []
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
                flops = float(n) ** 3 / (3.0 if what == "eliminate" else 1.0)
                res.append("%8.2f ms %6.1f" % (best * 1e3, flops / best / 1e12))
            print("n = %6d: %s | %s | %s" % (n, *res), flush=True)
            ctx.partial_destroy(pf)
            for q in (m, red, inv):
                q.close()
            ctx.block_destroy(0)


if __name__ == "__main__":
    main()
