#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
TAG=r03
T0=$SECONDS; timeout 900 python bench.py 2> $O/default_run.err | tail -1 > $O/${TAG}_bench_default_run.json; echo "python bench.py (no flags: cfg3, 2 timed steps + 1 warm-up, the one-chain step, the CPU baseline sample in both schedules): $((SECONDS - T0)) s wall clock" > $O/${TAG}_bench_default_run_time.txt
cp $O/${TAG}_bench_default_run.json $O/${TAG}_bench_cfg3.json
DNAGPU_MULTI_THREAD=0 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_one_chain.json
cut -c1-220 $O/${TAG}_bench_cfg3.json; cat $O/${TAG}_bench_default_run_time.txt
