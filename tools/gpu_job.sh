#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$SECONDS
timeout 1500 python -m pytest tests -q -m gpu -x --durations=10 > gpurun_out/gpu_suite_full.txt 2>&1
grep "s call" gpurun_out/gpu_suite_full.txt | head -10 > gpurun_out/gpu_suite.txt
grep "passed\|failed" gpurun_out/gpu_suite_full.txt | tail -1 >> gpurun_out/gpu_suite.txt
echo "suite: $((SECONDS - T0)) s" >> gpurun_out/gpu_suite.txt
tail -3 gpurun_out/gpu_suite.txt
grep -B2 -A12 "Error\|FAILED" gpurun_out/gpu_suite_full.txt | head -40
