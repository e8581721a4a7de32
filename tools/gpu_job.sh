#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
rm -rf gpurun_out/profiles_new
TAG=r02 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
echo "refresh rc=$?"
ls gpurun_out/profiles_new | wc -l
