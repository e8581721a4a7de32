#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -q -x -m gpu --durations=8 > gpurun_out/t_all.log 2>&1
echo "all rc=$?"; tail -n 14 gpurun_out/t_all.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_defer.json
DNAGPU_MULTI_THREAD=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_defer_one.json
python bench.py --workload cfg4_slice --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_slice_defer.json
python - <<'PY'
import json
for n in ("cfg3_defer", "cfg3_defer_one", "cfg4_slice_defer"):
    try:
        j = json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"].get("frac_end_to_end"), j.get("cholesky_tflops"), j["check"])
    except Exception as e:
        print(n, "ERR", e)
PY
