#!/bin/bash
R="$GRAFT_REPO_ROOT"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
for rows in -1 0; do
  if [ $rows -ge 0 ]; then export DNAGPU_TILE_ROWS=$rows; else unset DNAGPU_TILE_ROWS; fi
  echo "== rows=$rows one chain"
  DNAGPU_MULTI_THREAD=0 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['frac'])"
  echo "== rows=$rows four chains, kernel trace"
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt$rows -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/kt$rows.log 2>&1
  grep '^{"metric"' /tmp/kt$rows.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['frac'])"
  f=$(find /tmp/kt$rows -name '*kernel_stats.csv' | head -1)
  head -8 $f | cut -c1-200
done
