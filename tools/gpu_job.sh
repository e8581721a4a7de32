#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "chain_plan" 2>&1 | tail -15
echo "chain_runs 0: $(timeout 600 python bench.py --workload dnasegment150 --chain-runs 0 --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c100-240)"
echo "smallblocks chain_runs 0: $(timeout 600 python bench.py --workload smallblocks --chain-runs 0 --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c100-240)"
