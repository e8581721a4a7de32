#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/j11
timeout 900 python -m pytest tests/test_gpu_adjust.py tests/test_gpu_batch.py tests/test_gpu_distributed.py -q -m gpu -x --durations=5 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "dnasegment150 or smallblocks or cfg3" --durations=4 2>&1 | tail -8
for w in dnasegment150 smallblocks; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain > gpurun_out/j11/$w.json 2> gpurun_out/j11/$w.err
  cut -c1-250 gpurun_out/j11/$w.json; grep "phase" gpurun_out/j11/$w.err | tail -24
done
