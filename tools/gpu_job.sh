#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain $EXTRA 2>&1 | grep -E "ms_per_step|rror" | cut -c90-250; }
EXTRA="" run DNAGPU_SCHUR_SPLIT=0.2
EXTRA="" run DNAGPU_SCHUR_SPLIT=0.15
EXTRA="" run DNAGPU_SCHUR_SPLIT=0.1
EXTRA="" run DNAGPU_SCHUR_SPLIT=0.07
EXTRA="" run DNAGPU_SCHUR_SPLIT=0.25
echo "== default run"
T0=$SECONDS; timeout 900 python bench.py 2> gpurun_out/default_run.err | tail -1 > gpurun_out/r03_bench_default_run.json; echo "python bench.py (no flags: cfg3, 2 timed steps + 1 warm-up, the one-chain step, the CPU baseline sample in both schedules): $((SECONDS - T0)) s wall clock" | tee gpurun_out/r03_bench_default_run_time.txt
cut -c1-400 gpurun_out/r03_bench_default_run.json
