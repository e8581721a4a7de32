#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
TAG=r03 timeout 3000 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
echo "refresh rc=$?"; tail -5 gpurun_out/refresh.log; ls gpurun_out/profiles_new | grep r03 | head -50
