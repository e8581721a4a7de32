#!/bin/bash
# one gpurun job: golden record of the oracle (host cores) beside the GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
free -g | head -2 > gpurun_out/host.txt; nproc >> gpurun_out/host.txt
( ORACLE_THREADS=${ORACLE_THREADS:-64} timeout 2400 python tools/make_fullsize_golden.py cfg3 gpurun_out/cfg3_oracle.npz > gpurun_out/golden_cfg3.log 2>&1 ) &
GOLD=$!
timeout 900 python -m pytest tests/test_gpu_matrix.py -x -q > gpurun_out/t_matrix.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -k "not oracle_record" > gpurun_out/t_fullsize.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_matrix.py > gpurun_out/t_all.log 2>&1
wait $GOLD
tail -3 gpurun_out/t_matrix.log gpurun_out/t_fullsize.log gpurun_out/t_all.log gpurun_out/golden_cfg3.log
