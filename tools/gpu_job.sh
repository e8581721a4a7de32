#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
echo "== look-ahead + batch tests"
timeout 900 python -m pytest tests/test_gpu_lookahead.py tests/test_gpu_batch.py tests/test_gpu_matrix.py -x -q 2>&1 | tail -15
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain $EXTRA 2>&1 | grep -E "ms_per_step|rror" | cut -c90-250; }
EXTRA="" run DNAGPU_LOOKAHEAD=1
EXTRA="" run DNAGPU_LOOKAHEAD=0
EXTRA="" run DNAGPU_LOOKAHEAD=1 DNAGPU_MULTI_THREAD=0
EXTRA="" run DNAGPU_LOOKAHEAD=0 DNAGPU_MULTI_THREAD=0
EXTRA="" run DNAGPU_LOOKAHEAD=1 DNAGPU_BATCH=0 DNAGPU_MULTI_THREAD=0
EXTRA="" run DNAGPU_LOOKAHEAD=0 DNAGPU_BATCH=0 DNAGPU_MULTI_THREAD=0
EXTRA="--workload cfg2" run DNAGPU_LOOKAHEAD=1
EXTRA="--workload cfg2" run DNAGPU_LOOKAHEAD=0
EXTRA="--workload cfg2" run DNAGPU_LOOKAHEAD=1 DNAGPU_LOOKAHEAD_MIN_TILES=300
EXTRA="--stage" run DNAGPU_LOOKAHEAD=0
