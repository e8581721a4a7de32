#!/usr/bin/env python
"""Where the time of ONE matrix inverse goes: run under rocprofv3 --kernel-trace, this times a few dnagpu_invert calls of one order; called with
`summary <dir> <n>` it reads the trace back and splits the LAST inverse (first leaf launch to the LAUUM's end) into leaves, tile products by
launch shape, everything else, and the gaps between consecutive kernels.
  rocprofv3 --kernel-trace -d /tmp/kt_inv -o p --output-format csv -- python tools/gpu_inverse_trace.py run 2048
  python tools/gpu_inverse_trace.py summary /tmp/kt_inv 2048"""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(ns):
    import numpy as np
    from dynadjust_amd.device import DeviceContext
    from tools.gpu_inverse_bench import spd_packed
    n = 3 * ns
    ap = spd_packed(n, np.random.default_rng(1))
    with DeviceContext(0) as ctx:
        m = ctx.matrix(n)
        for rep in range(3):
            m.upload_packed(ap, n)
            ctx.sync()
            t0 = time.perf_counter()
            m.invert()
            ctx.sync()
            print("n = %d inverse %d: %.2f ms" % (n, rep, (time.perf_counter() - t0) * 1e3), flush=True)
        m.close()


def summary(d, ns):
    rows = []
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last inverse: from the last unpack_lower_kernel (the upload before it) to the end
    last_unpack = max(i for i, r in enumerate(rows) if "unpack_lower" in r[2])
    inv = [r for r in rows[last_unpack + 1:] if "pack_lower" not in r[2]]
    t0, t1 = inv[0][0], max(r[1] for r in inv)
    cls = {}
    def key(name):
        if "leaf_potrf" in name: return "leaf (128 x 128 potrf + trtri, one workgroup)"
        if "gemm_f64_dma" in name: return "tile products, 128-tile throughput kernel"
        if "gemm_f64_kernel" in name and ", 64, 4" in name: return "tile products on 64 x 64 tiles (launches of < 512 128-tiles)"
        if "gemm_f64_kernel" in name and ", 32, 4" in name: return "tile products on 32 x 32 tiles (launches of < 64 128-tiles)"
        return "other"
    gaps, end = 0, None
    for s, e, name in inv:
        k = cls.setdefault(key(name), [0, 0])
        k[0] += 1
        k[1] += e - s
        if end is not None and s > end:
            gaps += s - end
        end = e if end is None else max(end, e)
    total = t1 - t0
    n = 3 * ns
    print("# one inverse of n = %d (%d tiles) on 1 x MI355X, one chain, from a rocprofv3 --kernel-trace: %d kernels, %.2f ms from the first start to the last end"
          " (%.1f TFLOP/s at n^3)" % (n, (n + 127) // 128, len(inv), total / 1e6, n ** 3 / (total / 1e9) / 1e12))
    for name, (c, ns_) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        print("%-70s %5d launches %9.3f ms %6.1f %%   (%.1f us each)" % (name, c, ns_ / 1e6, 100.0 * ns_ / total, ns_ / 1e3 / c))
    print("%-70s %5d gaps     %9.3f ms %6.1f %%   (%.1f us each)" % ("between the end of a kernel and the start of the next", len(inv) - 1, gaps / 1e6, 100.0 * gaps / total,
                                                                 gaps / 1e3 / max(1, len(inv) - 1)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        summary(sys.argv[2], int(sys.argv[3]))
