#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adjust.py tests/test_gpu_distributed.py tests/test_gpu_terrestrial.py -q -x > gpurun_out/t_part.log 2>&1
echo "tests rc=$?"; tail -n 5 gpurun_out/t_part.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_cfg3_lend.json
python bench.py --workload cfg4_slice --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/cfg4.err | tail -1 > gpurun_out/bench_cfg4_slice_lend.json
python - <<'PY'
import json
for n in ("cfg3_lend", "cfg4_slice_lend"):
    try:
        j = json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, j["ms_per_step"], j["value"], j["roofline"]["frac"], j["config"].get("completions_per_step"), j["config"].get("solves_per_step"), j.get("cholesky_tflops"))
    except Exception as e:
        print(n, "ERR", e)
PY
tail -3 gpurun_out/cfg4.err
