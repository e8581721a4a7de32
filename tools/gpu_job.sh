#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain"
for i in 1 2 3; do
  echo "b32: $($B 2>/dev/null | cut -c100-215)"
  echo "b16: $(DNAGPU_LIB_OVERRIDE=$PWD/variants/libdnagpu_b16.so $B 2>/dev/null | cut -c100-215)"
done
