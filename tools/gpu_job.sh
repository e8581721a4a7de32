#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | grep -E "phase|ms_per_step|rror" | cut -c1-250 | tail -8; }
run DNAGPU_PHASE_TIMES=1
run A=1
run DNAGPU_MULTI_THREAD=0
