#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 400 python -m pytest tests/test_boundary_cpp.py -q -m gpu 2>&1 | tail -5
timeout 1500 python tools/make_fullsize_golden.py cfg3 gpurun_out/cfg3_oracle.npz > gpurun_out/cfg3_golden.log 2>&1
echo "golden rc=$?"; tail -3 gpurun_out/cfg3_golden.log | cut -c1-600; ls -la gpurun_out/cfg3_oracle.npz
