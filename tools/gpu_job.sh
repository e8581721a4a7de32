#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['ms_per_step'], j['roofline']['frac'])"; }
run base
DNAGPU_CHAINS=3 run chains3
DNAGPU_CHAINS=5 run chains5
DNAGPU_CHAINS=6 run chains6
DNAGPU_CHAINS=8 run chains8
DNAGPU_SMALL_TILES=256 run small256
DNAGPU_SMALL_TILES=1024 run small1024
