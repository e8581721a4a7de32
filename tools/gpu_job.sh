#!/bin/bash
R="$GRAFT_REPO_ROOT"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
(cd $R && timeout 600 python -m pytest tests/test_gpu_terrestrial.py -q -x -k "text_files" 2>&1 | tail -3)
timeout 900 rocprofv3 --kernel-trace -d /tmp/kt -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# the timed step = the second half of the big-kernel activity: take the last 32 LAUUM (<true, true, 4>) launches' span
lau = sorted([e for e in ev if "gemm_f64_dma_kernel<true, true" in e[2]])
step = lau[-32:]
t0 = min(e[0] for e in step) - 150_000_000     # (the step starts ~0.1 s before its first LAUUM: condensing)
t1 = max(e[1] for e in ev)
def union(pred):
    iv = sorted((max(s, t0), min(e, t1)) for s, e, n in ev if pred(n) and e > t0 and s < t1)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot / 1e9
big = lambda n: "gemm_f64_dma_kernel" in n
print("window %.3f s" % ((t1 - t0) / 1e9))
print("some 128-tile GEMM running   %.3f s" % union(big))
print("some GEMM (any tile) running %.3f s" % union(lambda n: "gemm_f64" in n))
print("some kernel running          %.3f s" % union(lambda n: True))
# concurrency histogram of the 128-tile GEMMs
pts = []
for s, e, n in ev:
    if big(n) and e > t0 and s < t1:
        pts.append((max(s, t0), 1)); pts.append((min(e, t1), -1))
pts.sort()
level, last, hist = 0, t0, {}
for t, d in pts:
    hist[level] = hist.get(level, 0) + (t - last); last = t; level += d
hist[level] = hist.get(level, 0) + (t1 - last)
print("128-tile GEMMs in flight: " + ", ".join("%d: %.3f s" % (k, v / 1e9) for k, v in sorted(hist.items())))
PY
