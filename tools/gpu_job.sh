#!/bin/bash
# the job of the moment for `gpurun -- bash tools/gpu_job.sh` (edited per measurement; this is the round-end check)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/t_all.log 2>&1
echo "all rc=$?" > gpurun_out/job.status
tail -n 6 gpurun_out/t_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/job.status
tail -2 gpurun_out/smoke.log; cat gpurun_out/job.status
