#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j5
timeout 900 python -m pytest tests/test_gpu_adjust.py -q -m gpu -x -k "lock_step or singular or many_small or oscill or suspect" 2>&1 | tail -5
for w in dnasegment150; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain > gpurun_out/j5/$w.json 2> gpurun_out/j5/$w.err
  cut -c1-250 gpurun_out/j5/$w.json; grep "phase" gpurun_out/j5/$w.err | tail -24 | grep "plan\|iteration 1\|iteration 2\|variance\|AdjustNetwork"
done
