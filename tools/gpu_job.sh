#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
TAG=r03
cd /tmp && export TMPDIR=/tmp
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/pmc_$c.log 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python $R/tools/rocprof_summary.py pmc /tmp/pmc_$c $O/${TAG}_cfg3_pmc_$lc.txt "rocprofv3 --pmc $c --kernel-trace -- $CMD   (${TAG}, cfg3)"
done
(cd $O && python $R/tools/pmc_traffic_json.py ${TAG}_cfg3_pmc_fetch_size.txt ${TAG}_cfg3_pmc_write_size.txt ${TAG}_hbm_traffic.json cfg3 > /dev/null)
cp $O/${TAG}_hbm_traffic.json $R/profiles/
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/kt.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/kt $O/${TAG}_cfg3_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $CMD   (${TAG}, cfg3: 100 172 stations / 16 blocks, condensed schedule, batched blocks, four chains, 1 x MI355X)"
DNAGPU_MULTI_THREAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/kt1.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/kt1 $O/${TAG}_cfg3_kernel_stats_one_chain.txt "DNAGPU_MULTI_THREAD=0 rocprofv3 --kernel-trace --stats -- $CMD   (${TAG}, cfg3, batched blocks, ONE chain: kernel durations without overlap)"
cd $R
timeout 600 python bench.py --steps 2 --warmup 1 2> $O/bench_cfg3.err | tail -1 > $O/${TAG}_bench_cfg3.json
DNAGPU_MULTI_THREAD=0 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_one_chain.json
DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | grep "^\[phase\]" | tail -7 > $O/${TAG}_cfg3_phase_times.txt
DNAGPU_MULTI_THREAD=0 DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | grep "^\[phase\]" | tail -7 > $O/${TAG}_cfg3_phase_times_one_chain.txt
T0=$SECONDS; timeout 900 python bench.py 2> $O/default_run.err | tail -1 > $O/${TAG}_bench_default_run.json; echo "python bench.py (no flags: cfg3, 2 timed steps + 1 warm-up, the one-chain step, the CPU baseline sample in both schedules): $((SECONDS - T0)) s wall clock" > $O/${TAG}_bench_default_run_time.txt
cut -c1-200 $O/${TAG}_bench_cfg3.json; cat $O/${TAG}_bench_default_run_time.txt
