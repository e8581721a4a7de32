#!/bin/bash
# TAG=r0N bash tools/refresh_profiles.sh
# Regenerates the round's measurement records under gpurun_out/profiles_new/ on the GPU box (run through gpurun from the repo root): bench
# records, rocprofv3 kernel stats, PMC passes (MFMA utilisation, HBM traffic, L2 hit rates: separate passes), inverse rates, the rank-share model.
R=$PWD
TAG=${TAG:-r06}
O=$R/gpurun_out/profiles_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
# ---- bench records ----
T0=$SECONDS; timeout 900 $B 2> $O/default_run.err | tail -1 > $O/${TAG}_bench_default_run.json; echo "python bench.py (no flags: cfg3, one-chain step, the step without factor reuse, CPU baseline sample): $((SECONDS - T0)) s wall clock" > $O/${TAG}_bench_default_run_time.txt
timeout 600 $B --steps 3 --warmup 1 --no-cpu-baseline 2> $O/bench_cfg3.err | tail -1 > $O/${TAG}_bench_cfg3.json
timeout 600 $B --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg2.json
timeout 600 $B --workload smallblocks --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_smallblocks.json
timeout 900 $B --workload dnasegment150 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150.json
timeout 600 $B --workload dnasegment150 --chain-runs 0 --steps 3 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150_chains_step_by_step.json
timeout 600 $B --workload smallblocks --chain-runs 0 --steps 3 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg 2>/dev/null | tail -1 > $O/${TAG}_bench_smallblocks_chains_step_by_step.json
timeout 600 $B --workload cfg3_ragged --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_ragged.json
timeout 1500 $B --workload dnasegment150_10x --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150_10x.json
timeout 600 $B --workload cfg4_slice --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg4_slice.json
timeout 600 $B --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg --variance-propagation 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_variance_propagation.json
timeout 600 $B --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg --stage 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_staged.json
timeout 600 $B --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg --reference-schedule 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_reference_schedule.json
DNAGPU_FORCE_DISTRIBUTED=1 timeout 600 $B --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_rccl_one_rank.json
for w in cfg3 smallblocks dnasegment150 dnasegment150_10x; do
  DNAGPU_PHASE_TIMES=1 timeout 600 $B --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg 2>&1 | grep "^\[phase\]" | tail -40 > $O/${TAG}_${w}_phase_times.txt
done
# ---- kernel traces ----
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-refactor-leg"
F="--steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-refactor-leg"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- timeout 600 $B $F > $O/kt.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/kt $O/${TAG}_cfg3_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $CMD   (${TAG}, cfg3: 100 172 stations / 16 blocks, condensed schedule, factor reuse, four chains, 1 x MI355X)"
grep '^{"metric"' $O/kt.log | tail -1 > $O/${TAG}_bench_cfg3_profiled_step.json
DNAGPU_MULTI_THREAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o p --output-format csv -- timeout 600 $B $F > $O/kt1.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/kt1 $O/${TAG}_cfg3_kernel_stats_one_chain.txt "DNAGPU_MULTI_THREAD=0 rocprofv3 --kernel-trace --stats -- $CMD   (${TAG}, cfg3, ONE chain: kernel durations without overlap)"
for w in smallblocks dnasegment150; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o p --output-format csv -- timeout 600 $B --workload $w $F > $O/kt_$w.log 2>&1
  python $R/tools/rocprof_summary.py stats /tmp/kt_$w $O/${TAG}_${w}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w ${F}   (${TAG}; the trace covers PrepareAdjustment, ONE adjustment and the closing statistics)"
done
# ---- PMC passes (one chain; each counter set its own run) ----
if [ -z "$SKIP_PMC" ]; then
DNAGPU_MULTI_THREAD=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_mfma -o p --output-format csv -- timeout 600 $B $F > $O/pmc_mfma.log 2>&1
python $R/tools/rocprof_summary.py pmc /tmp/pmc_mfma $O/${TAG}_cfg3_pmc_mfma_util.txt "DNAGPU_MULTI_THREAD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- $CMD   (${TAG}, cfg3, one chain: MFMA pipe utilisation per kernel)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- timeout 600 $B $F > $O/pmc_$c.log 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python $R/tools/rocprof_summary.py pmc /tmp/pmc_$c $O/${TAG}_cfg3_pmc_$lc.txt "rocprofv3 --pmc $c --kernel-trace -- $CMD   (${TAG}, cfg3)"
done
(cd $O && python $R/tools/pmc_traffic_json.py ${TAG}_cfg3_pmc_fetch_size.txt ${TAG}_cfg3_pmc_write_size.txt ${TAG}_hbm_traffic.json cfg3 > /dev/null)
DNAGPU_MULTI_THREAD=0 timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/pmc_l2 -o p --output-format csv -- timeout 600 $B $F > $O/pmc_l2.log 2>&1
python $R/tools/rocprof_summary.py pmc /tmp/pmc_l2 $O/${TAG}_cfg3_pmc_l2_hits.txt "DNAGPU_MULTI_THREAD=0 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -- $CMD   (${TAG}, cfg3, one chain: L2 hits / misses per kernel; hit rate = HIT / (HIT + MISS))"
fi
# ---- rates and the N-GPU model ----
{ echo "# python tools/gpu_inverse_bench.py on 1 x MI355X (${TAG}), through the C-ABI, best of 3 timed repetitions, one chain"; timeout 300 python $R/tools/gpu_inverse_bench.py 2>/dev/null; } > $O/${TAG}_inverse_rates.txt
timeout 600 python $R/tools/gpu_rank_share.py > $O/${TAG}_rank_share.txt 2>/dev/null
# cfg4 / cfg5 at full size on one GPU (7 min each): TAG=cfg4 bash tools/run_cfg4_1gpu.sh ; TAG=cfg5 bash tools/run_cfg4_1gpu.sh --variance-propagation
ls -la $O
