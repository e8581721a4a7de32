#!/bin/bash
# round-5 job 1: factor reuse -- the tests that exercise it, then cfg3 / smallblocks with phase times
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/j1
timeout 900 python -m pytest tests/test_gpu_adjust.py tests/test_gpu_batch.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/j1/tests.txt
cat gpurun_out/j1/tests.txt
DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/j1/cfg3.json 2> gpurun_out/j1/cfg3.err
grep "phase" gpurun_out/j1/cfg3.err | tail -12; tail -c 600 gpurun_out/j1/cfg3.json
DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload smallblocks --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/j1/smallblocks.json 2> gpurun_out/j1/smallblocks.err
grep "phase" gpurun_out/j1/smallblocks.err | tail -12; tail -c 600 gpurun_out/j1/smallblocks.json
