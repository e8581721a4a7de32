#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -q -x -m gpu --durations=5 > gpurun_out/t_all.log 2>&1
echo "all rc=$?"; tail -n 12 gpurun_out/t_all.log
for dv in 2 1; do
DNAGPU_DEFER_VARIANCES=$dv python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_dv$dv.json
DNAGPU_DEFER_VARIANCES=$dv DNAGPU_MULTI_THREAD=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_one_dv$dv.json
done
python bench.py --workload cfg4_slice --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg4_slice_dv2.json
python - <<'PY'
import json
for n in ("cfg3_dv2", "cfg3_one_dv2", "cfg3_dv1", "cfg3_one_dv1", "cfg4_slice_dv2"):
    try:
        j = json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, j["ms_per_step"], j["value"], j["roofline"]["frac"], j["roofline"].get("frac_end_to_end"), j.get("cholesky_tflops"), j["check"]["sigma_zero"], j["check"]["max_abs_error_vs_truth_m"])
    except Exception as e:
        print(n, "ERR", e)
PY
