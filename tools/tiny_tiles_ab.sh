#!/bin/bash
# A/B of the 32 x 32 block tiles for tiny launches (DNAGPU_TINY_TILES, default 16; 0 = off) on the GPU box: python tools ... > gpurun_out/r04/tiny_tiles.txt
R=$PWD
for t in 0 8 16 32 64; do
  echo "== DNAGPU_TINY_TILES=$t"
  echo "-- single matrices (tools/gpu_inverse_bench.py)"
  DNAGPU_TINY_TILES=$t timeout 300 python $R/tools/gpu_inverse_bench.py 2>/dev/null
  for w in smallblocks cfg3; do
    DNAGPU_TINY_TILES=$t timeout 600 python $R/bench.py --workload $w --no-cpu-baseline --no-one-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- $w', round(d['value']), 'stations/s', round(d['ms_per_step'],1), 'ms per adjustment, frac', round(d['roofline']['frac'],4))"
  done
done
echo "== cfg2, cfg3 one chain (default / off)"
for t in 16 0; do
  DNAGPU_TINY_TILES=$t timeout 600 python $R/bench.py --workload cfg2 --steps 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- cfg2 tiny=$t', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],4))"
  DNAGPU_MULTI_THREAD=0 DNAGPU_TINY_TILES=$t timeout 600 python $R/bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- cfg3 one chain tiny=$t', round(d['value']), round(d['ms_per_step'],1), round(d['roofline']['frac'],4))"
done
