#!/usr/bin/env python
"""How busy the device is during the lock-step chain phase: reads a rocprofv3 --kernel-trace directory of `bench.py --workload <w> --steps 1
--warmup 0` and looks at the window from the first cb_assemble_kernel (the first batch of the chain plan) to the first small_solve_kernel after
it (the rigorous solves of iteration 1): kernels, summed duration, time with at least one kernel executing, per stream (queue)."""
import csv, glob, os, sys
from collections import defaultdict
rows = []
for path in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
t0 = next(s for s, e, n, q in rows if "cb_assemble_kernel" in n)
t1 = next(s for s, e, n, q in rows if s > t0 and "small_solve_kernel" in n)
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
busy, end = 0, None
for s, e, n, q in win:
    if end is None or s >= end:
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
tot = sum(e - s for s, e, n, q in win)
print("# chain phase of iteration 1: %.2f ms, %d kernels, summed duration %.2f ms, some kernel executing %.2f ms (%.0f %%), average overlap %.2f"
      % ((t1 - t0) / 1e6, len(win), tot / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / max(busy, 1)))
perq = defaultdict(lambda: [0, 0])
for s, e, n, q in win:
    perq[q][0] += 1
    perq[q][1] += e - s
for q, (c, d) in sorted(perq.items()):
    print("#   queue %s: %d kernels, %.2f ms" % (q, c, d / 1e6))
perk = defaultdict(lambda: [0, 0])
for s, e, n, q in win:
    perk[n.split("(")[0][:70]][0] += 1
    perk[n.split("(")[0][:70]][1] += e - s
for n, (c, d) in sorted(perk.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-72s %6d %9.2f ms %7.1f us each" % (n, c, d / 1e6, d / 1e3 / c))
