#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j9
B="timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain"
for s in 0.12 0.16 0.25; do echo "split $s: $(DNAGPU_XSPLIT=$s $B 2>/dev/null | cut -c100-240)"; done
for t in 256 1024; do echo "small_tiles $t: $(DNAGPU_BENCH_SMALL_TILES=$t $B 2>/dev/null | cut -c100-240)"; done
for r in 8 16 24 48 64; do echo "chain_runs $r: $(timeout 600 python bench.py --workload dnasegment150 --chain-runs $r --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c100-240)"; done
for r in 4 8 24; do echo "smallblocks chain_runs $r: $(timeout 600 python bench.py --workload smallblocks --chain-runs $r --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c100-240)"; done
