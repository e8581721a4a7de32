#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$SECONDS
timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -20 > gpurun_out/gpu_suite.txt
echo "suite: $((SECONDS - T0)) s" >> gpurun_out/gpu_suite.txt
tail -14 gpurun_out/gpu_suite.txt
TAG=r05 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
echo "total: $((SECONDS - T0)) s"
for f in gpurun_out/profiles_new/r05_bench_*.json; do echo $f; cut -c100-260 $f; echo; done
cat gpurun_out/profiles_new/r05_inverse_rates.txt
cat gpurun_out/profiles_new/r05_hbm_traffic.json | head -30
