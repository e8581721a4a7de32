#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j8
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_exact.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_adjust.py -q -m gpu -x -k "lock_step or many_small or deserialise or statistics or reuse" 2>&1 | tail -3
for w in dnasegment150 smallblocks; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain > gpurun_out/j8/$w.json 2> gpurun_out/j8/$w.err
  cut -c1-250 gpurun_out/j8/$w.json; grep "phase" gpurun_out/j8/$w.err | tail -24 | grep "variance\|AdjustNetwork" | tail -2
done
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "record" 2>&1 | tail -3
