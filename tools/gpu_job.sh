#!/bin/bash
# one gpurun job (edit per need): full suite, smoke, default bench, profile refresh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/t_all.log 2>&1
echo "all rc=$?" > gpurun_out/job.status
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/job.status
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench rc=$?" >> gpurun_out/job.status
TAG=r02 timeout 2400 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
echo "refresh rc=$?" >> gpurun_out/job.status
tail -n 4 gpurun_out/t_all.log gpurun_out/smoke.log
cat gpurun_out/job.status
