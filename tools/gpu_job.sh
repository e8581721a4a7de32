#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['hbm_gb'], d['config']['batched_block_steps_per_step'])"
timeout 400 python bench.py --workload cfg4_slice --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['hbm_gb'], d['config']['batched_block_steps_per_step'])"
