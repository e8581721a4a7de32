#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
TAG=r02 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
echo "refresh rc=$?" > gpurun_out/job.status
tail -5 gpurun_out/refresh.log
