#!/bin/bash
# second pass of tools/tiny_tiles_ab.sh: larger thresholds
R=$PWD
for t in 64 128 256 512; do
  echo "== DNAGPU_TINY_TILES=$t"
  DNAGPU_TINY_TILES=$t timeout 300 python $R/tools/gpu_inverse_bench.py 2>/dev/null | grep "n = " | head -4
  for w in smallblocks cfg3 cfg2; do
    DNAGPU_TINY_TILES=$t timeout 600 python $R/bench.py --workload $w --no-cpu-baseline --no-one-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- $w', round(d['value']), 'stations/s', round(d['ms_per_step'],1), 'ms per adjustment, frac', round(d['roofline']['frac'],4))"
  done
  DNAGPU_MULTI_THREAD=0 DNAGPU_TINY_TILES=$t timeout 600 python $R/bench.py --no-cpu-baseline --no-one-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- cfg3 one chain', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],4))"
done
