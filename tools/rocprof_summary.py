#!/usr/bin/env python
"""Summaries of rocprofv3 output directories, written as the text files kept under profiles/.

  python tools/rocprof_summary.py stats <dir> <out.txt> "<header>"     kernel-trace CSV -> per-kernel calls / total / avg / pct
  python tools/rocprof_summary.py pmc   <dir> <out.txt> "<header>"     counter_collection CSV -> per-kernel counter sums

PMC notes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads, so hbm_read_bytes = 2 * FETCH_SIZE * 1024.  The two counters need separate passes.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def _find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits


def stats(d, out, header):
    per = defaultdict(lambda: [0, 0.0])
    t0, t1 = None, None
    n = 0
    gemm_iv, all_iv = [], []
    for path in _find(d, "kernel_trace.csv"):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
                k = per[row["Kernel_Name"]]
                k[0] += 1
                k[1] += (e - s) / 1e3
                t0 = s if t0 is None else min(t0, s)
                t1 = e if t1 is None else max(t1, e)
                n += 1
                all_iv.append((s, e))
                if "gemm_f64" in row["Kernel_Name"]:
                    gemm_iv.append((s, e))
    tot = sum(v[1] for v in per.values())
    # time during which at least one gemm kernel was executing (kernels of two streams overlap in multi-thread mode);
    # this is what bench.py reports as roofline.gemm_ms_per_step
    gemm_iv.sort()
    union, end = 0, None
    for s, e in gemm_iv:
        if end is None or s >= end:
            union += e - s
            end = e
        elif e > end:
            union += e - end
            end = e
    with open(out, "w") as f:
        f.write("# %s\n" % header)
        f.write("# kernel dispatches: %d, first start -> last end: %.3f s   (durations in microseconds)\n" % (n, (t1 - t0) / 1e9))
        f.write("# gemm_f64* kernels: %d dispatches, summed duration %.3f s, union of their intervals %.3f s\n"
                % (len(gemm_iv), sum(e - s for s, e in gemm_iv) / 1e9, union / 1e9))
        all_iv.sort()
        busy, end = 0, None
        for s, e in all_iv:
            if end is None or s >= end:
                busy += e - s
                end = e
            elif e > end:
                busy += e - end
                end = e
        f.write("# all kernels: summed duration %.3f s, time with at least one kernel executing %.3f s (average overlap %.2f), device idle %.3f s\n"
                % (tot / 1e6, busy / 1e9, (tot / 1e6) / max(busy / 1e9, 1e-12), (t1 - t0 - busy) / 1e9))
        f.write("%-112s %7s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, (c, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            f.write("%-112s %7d %14.1f %12.3f %7.2f\n" % (name[:112], c, us, us / c, 100.0 * us / tot))


def pmc(d, out, header):
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for path in _find(d, "counter_collection.csv"):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                per[name][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[name].add((path, row["Dispatch_Id"]))
    counters = sorted({c for v in per.values() for c in v})
    with open(out, "w") as f:
        f.write("# %s\n" % header)
        f.write("# per kernel: dispatches, then for every counter the sum over dispatches and the mean per dispatch\n")
        for name, vals in sorted(per.items(), key=lambda kv: -sum(kv[1].values())):
            n = len(calls[name])
            f.write("%s\n    dispatches %d\n" % (name[:140], n))
            for c in counters:
                if c in vals:
                    f.write("    %-14s sum %.6e   per dispatch %.6e\n" % (c, vals[c], vals[c] / n))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and vals.get("GRBM_GUI_ACTIVE", 0) > 0:
                # rocprofv3's own derived metric: MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM).  The csv
                # rows of GRBM_GUI_ACTIVE are per XCD and summed here, hence / 8; 1024 SIMDs (256 CU x 4)
                f.write("    MfmaUtil       %.1f %%   (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD x 1024 SIMD))\n" %
                        (100.0 * vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))


if __name__ == "__main__":
    if len(sys.argv) != 5 or sys.argv[1] not in ("stats", "pmc"):
        raise SystemExit(__doc__)
    (stats if sys.argv[1] == "stats" else pmc)(sys.argv[2], sys.argv[3], sys.argv[4])
