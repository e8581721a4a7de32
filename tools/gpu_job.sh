#!/bin/bash
R="$GRAFT_REPO_ROOT"
mkdir -p $R/gpurun_out
cd $R
export GPU_MAX_HW_QUEUES=24
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/err.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['ms_per_step'], j['roofline']['frac'])" || tail -5 gpurun_out/err.log; }
run base
DNAGPU_BLOCKING_STREAMS=1 run blocking
