#!/usr/bin/env python
"""Reads the per-task clocks a DAG launch left (DNAGPU_DAG_TRACE=<prefix>, sym_inverse.hip run_dag) and says where the time went:
python tools/dag_trace_report.py <file.bin> [...]"""
import sys
import numpy as np

TASK_DT = np.dtype([("a_off", "<u4"), ("b_off", "<u4"), ("c_off", "<u4"), ("it", "<u2"), ("jt", "<u2"), ("kb", "<u2"), ("ke", "<u2"),
                    ("type", "u1"), ("bufs", "u1"), ("flags", "u1"), ("pad", "u1"), ("flag", "<u4"), ("dep0", "<u4"), ("ndep", "<u4")])


def report(path):
    raw = open(path, "rb").read()
    hdr = np.frombuffer(raw[:64], dtype="<u8")
    n, kind, ti, tj, what, tsz = (int(x) for x in hdr[:6])
    tr = np.frombuffer(raw[64:64 + 32 * n], dtype="<u8").reshape(n, 4).astype(np.float64)
    tasks = np.frombuffer(raw[64 + 32 * n:64 + 32 * n + tsz * n], dtype=TASK_DT)
    t0 = tr[:, 0].min()
    start, ready, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0      # microseconds
    span = end.max()
    body_end = (tr[:, 3] - t0) / 100.0
    wait, run = ready - start, end - ready
    notify = end - body_end
    print(f"{path}: kind {kind} ti {ti} tj {tj} what {what}: {n} tasks, makespan {span / 1e3:.2f} ms, sum run {run.sum() / 1e3:.1f} ms, sum wait {wait.sum() / 1e3:.1f} ms, "
          f"slot-time {(end - start).sum() / 1e3:.1f} ms = {(end - start).sum() / span:.0f} slots on average, running {run.sum() / span:.0f}")
    nk = (tasks["ke"].astype(int) - tasks["kb"].astype(int))
    for ty, name in ((0, "NT128"), (1, "NN128"), (2, "TN128"), (3, "leaf"), (4, "NT64"), (5, "NN64"), (6, "TN64")):
        m = tasks["type"] == ty
        if not m.any():
            continue
        line = f"  {name:6s} {m.sum():7d} tasks, run {run[m].sum() / 1e3:9.1f} ms (of it release + notify {notify[m].sum() / 1e3:8.1f} ms, median {np.median(notify[m]):6.1f} us, max {notify[m].max():7.1f}), wait {wait[m].sum() / 1e3:9.1f} ms"
        if ty != 3:
            ks = np.unique(nk[m])
            pick = ks[[0, len(ks) // 4, len(ks) // 2, -1]] if len(ks) > 4 else ks
            line += "; us per task at nk = " + ", ".join(f"{k}: {np.median(run[m & (nk == k)]):.1f}" for k in np.unique(pick))
            tot_k = nk[m].sum()
            line += f"; {run[m].sum() / max(1, tot_k):.2f} us per k-tile overall"
        else:
            line += f"; median {np.median(run[m]):.1f} us"
        print(line)
    # occupancy over time: running workgroups in 20 buckets
    nb = 20
    edges = np.linspace(0, span, nb + 1)
    occ = np.zeros(nb)
    wt = np.zeros(nb)
    for b in range(nb):
        lo, hi = edges[b], edges[b + 1]
        occ[b] = np.clip(np.minimum(end, hi) - np.maximum(ready, lo), 0, None).sum() / (hi - lo)
        wt[b] = np.clip(np.minimum(ready, hi) - np.maximum(start, lo), 0, None).sum() / (hi - lo)
    print("  running workgroups per 5 % of the makespan: " + " ".join(f"{x:.0f}" for x in occ))
    print("  waiting workgroups per 5 % of the makespan: " + " ".join(f"{x:.0f}" for x in wt))


for p in sys.argv[1:]:
    report(p)
