#!/bin/bash
# item: fabric traffic of the tile GEMM with the pair walk, resident and staged
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pairs
export GPU_MAX_HW_QUEUES=16
O=$GRAFT_REPO_ROOT/gpurun_out/pairs
R=$GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain $EXTRA 2>&1 | grep -E "ms_per_step|rror" | cut -c90-250; }
for P in 0 2000 6000; do
  EXTRA="" run DNAGPU_PAIR_TILES=$P
  EXTRA="--stage" run DNAGPU_PAIR_TILES=$P
done
cd /tmp && export TMPDIR=/tmp
for P in 0 2000; do
  DNAGPU_PAIR_TILES=$P timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f_$P -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/pmc_f_$P.log 2>&1
  python $R/tools/rocprof_summary.py pmc /tmp/pmc_f_$P $O/pairs_${P}_fetch_size.txt "DNAGPU_PAIR_TILES=$P rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 0   (r03, cfg3, batched blocks)"
  head -12 $O/pairs_${P}_fetch_size.txt
done
