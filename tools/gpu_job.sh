#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -q -x -m gpu --durations=12 > gpurun_out/t_all.log 2>&1
echo "all rc=$?" > gpurun_out/job.status
tail -n 20 gpurun_out/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/job.status
tail -3 gpurun_out/smoke.log; cat gpurun_out/job.status
