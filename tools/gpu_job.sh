#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for w in dnasegment150 smallblocks dnasegment150_10x cfg3; do
  echo "$w: $(DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c100-230)"
done
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_matrix.py -q -m gpu -x 2>&1 | tail -3
