#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
for w in 0 2 3 5 8; do
  echo "== DNAGPU_SPLIT_WINDOW=$w"
  DNAGPU_SPLIT_WINDOW=$w python tools/gpu_inverse_bench.py 2>/dev/null | tail -4
done
for w in 0 3 5; do
  DNAGPU_SPLIT_WINDOW=$w python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('window $w four chains', j['ms_per_step'], j['roofline']['frac'])"
  DNAGPU_SPLIT_WINDOW=$w DNAGPU_MULTI_THREAD=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('window $w one chain', j['ms_per_step'], j['roofline']['frac'])"
done
