#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j7
timeout 900 python -m pytest tests/test_gpu_adjust.py tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -5
for w in dnasegment150 smallblocks cfg3; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain > gpurun_out/j7/$w.json 2> gpurun_out/j7/$w.err
  cut -c1-250 gpurun_out/j7/$w.json; grep "phase" gpurun_out/j7/$w.err | tail -24 | grep "plan\|iteration 1\|iteration 2\|variance\|AdjustNetwork"
done
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "dnasegment150 or smallblocks or record" 2>&1 | tail -3
