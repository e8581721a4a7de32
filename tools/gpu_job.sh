#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
DNAGPU_POISON_ALLOC=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adjust.py tests/test_gpu_terrestrial.py -q -m gpu -x 2>&1 | tail -3
for w in dnasegment150 dnasegment150_10x cfg3; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>&1 | grep "PrepareAdjustment\|\"metric\"" | cut -c1-230
done
