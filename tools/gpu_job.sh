#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 16 24 32 40 48 64; do echo "dnasegment150 chain_runs $r: $(timeout 600 python bench.py --workload dnasegment150 --chain-runs $r --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c179-215)"; done
for r in 8 16 24 32; do echo "smallblocks chain_runs $r: $(timeout 600 python bench.py --workload smallblocks --chain-runs $r --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c178-215)"; done
for r in 40 60 96 128; do echo "10x chain_runs $r: $(timeout 900 python bench.py --workload dnasegment150_10x --chain-runs $r --steps 2 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c179-215)"; done
