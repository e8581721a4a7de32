#!/bin/bash
# sweep of the 64-tile threshold (DNAGPU_SMALL_TILES, default 512) with the 32-tile launches in place
R=$PWD
for t in 256 512 768 1024; do
  echo "== DNAGPU_SMALL_TILES=$t"
  DNAGPU_SMALL_TILES=$t timeout 300 python $R/tools/gpu_inverse_bench.py 2>/dev/null | grep "n = " | head -4
  for w in cfg3 cfg2; do
    DNAGPU_SMALL_TILES=$t timeout 600 python $R/bench.py --workload $w --no-cpu-baseline --no-one-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- $w', round(d['value']), 'stations/s', round(d['ms_per_step'],1), 'ms, frac', round(d['roofline']['frac'],4))"
  done
  DNAGPU_MULTI_THREAD=0 DNAGPU_SMALL_TILES=$t timeout 600 python $R/bench.py --no-cpu-baseline --no-one-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- cfg3 one chain', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],4))"
done
