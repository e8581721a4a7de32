#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/j9
timeout 600 python -m pytest tests/test_gpu_adjust.py -q -m gpu -x --durations=5 2>&1 | tail -12
for w in dnasegment150; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg > gpurun_out/j9/$w.json 2> gpurun_out/j9/$w.err
  cut -c1-250 gpurun_out/j9/$w.json; grep "phase" gpurun_out/j9/$w.err | grep "Reset\|AdjustNetwork" | tail -6
  for n in 2 4 8; do
    DNAGPU_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus $n --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg > gpurun_out/j9/${w}_share$n.json 2> gpurun_out/j9/${w}_share$n.err
    cut -c1-250 gpurun_out/j9/${w}_share$n.json; tail -3 gpurun_out/j9/${w}_share$n.err | cut -c1-300
  done
done
