#!/bin/bash
# TAG=r0N bash tools/refresh_profiles.sh
# Regenerates everything under gpurun_out/profiles_new/ on the GPU box (run through gpurun from the repo root):
# bench records, rocprofv3 kernel stats, PMC HBM traffic passes, GEMM variant table, MFMA probes.
R=$PWD
TAG=${TAG:-r03}          # profiles are named per round
O=$R/gpurun_out/profiles_new
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 2 --warmup 1 2> $O/bench_cfg3.err | tail -1 > $O/${TAG}_bench_cfg3.json
timeout 600 python $R/bench.py --workload cfg2 --steps 3 --warmup 1 2> $O/bench_cfg2.err | tail -1 > $O/${TAG}_bench_cfg2.json
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/kt.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/kt $O/${TAG}_cfg3_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $CMD   (${TAG}, cfg3: 100 172 stations / 16 blocks, condensed schedule, four chains, 1 x MI355X)"
DNAGPU_MULTI_THREAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/kt1.log 2>&1
python $R/tools/rocprof_summary.py stats /tmp/kt1 $O/${TAG}_cfg3_kernel_stats_one_chain.txt "DNAGPU_MULTI_THREAD=0 rocprofv3 --kernel-trace --stats -- $CMD   (${TAG}, cfg3, ONE chain: kernel durations without overlap)"
DNAGPU_MULTI_THREAD=0 timeout 600 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_one_chain.json   # (the profiled run above has no warm-up: first-touch allocations inside)
DNAGPU_BATCH=0 timeout 600 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_unbatched.json
DNAGPU_PHASE_TIMES=1 timeout 600 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | grep "^\[phase\]" | tail -7 > $O/${TAG}_cfg3_phase_times.txt
DNAGPU_MULTI_THREAD=0 DNAGPU_PHASE_TIMES=1 timeout 600 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | grep "^\[phase\]" | tail -7 > $O/${TAG}_cfg3_phase_times_one_chain.txt
T0=$SECONDS; timeout 900 python $R/bench.py 2> $O/default_run.err | tail -1 > $O/${TAG}_bench_default_run.json; echo "python bench.py (no flags: cfg3, CPU baseline sample, one-chain step): $((SECONDS - T0)) s wall clock" > $O/${TAG}_bench_default_run_time.txt
timeout 600 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain --reuse-inverses 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_reuse_inverses.json
timeout 600 python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain --reference-schedule 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_reference_schedule.json
timeout 600 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain --variances-every-iteration 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_variances_every_iteration.json
timeout 600 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain --stage 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_staged.json
timeout 600 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain --variance-propagation 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_variance_propagation.json
DNAGPU_FORCE_DISTRIBUTED=1 timeout 600 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg3_rccl_one_rank.json
timeout 600 python $R/bench.py --workload cfg4_slice --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain 2>/dev/null | tail -1 > $O/${TAG}_bench_cfg4_slice.json
grep '^{"metric"' $O/kt.log | tail -1 > $O/${TAG}_bench_cfg3_profiled_step.json
DNAGPU_MULTI_THREAD=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_mfma -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/pmc_mfma.log 2>&1
python $R/tools/rocprof_summary.py pmc /tmp/pmc_mfma $O/${TAG}_cfg3_pmc_mfma_util.txt "DNAGPU_MULTI_THREAD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- $CMD   (${TAG}, cfg3, one chain: MFMA pipe utilisation per kernel)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/pmc_$c.log 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python $R/tools/rocprof_summary.py pmc /tmp/pmc_$c $O/${TAG}_cfg3_pmc_$lc.txt "rocprofv3 --pmc $c --kernel-trace -- $CMD   (${TAG}, cfg3)"
done
(cd $O && python $R/tools/pmc_traffic_json.py ${TAG}_cfg3_pmc_fetch_size.txt ${TAG}_cfg3_pmc_write_size.txt ${TAG}_hbm_traffic.json cfg3 > /dev/null)
{ echo "# python tools/gpu_gemm_bench.py on 1 x MI355X (${TAG}): per-variant throughput of the fp64 tile GEMM, HIP-event timed, 3 launches each"
  echo "# fp64 MFMA peak 78.6 TFLOP/s; operands pseudo-random full-range mantissas"
  for v in dma4 dma8 reg4 reg8; do echo "== DNAGPU_GEMM_VARIANT=$v"; DNAGPU_GEMM_VARIANT=$v timeout 300 python $R/tools/gpu_gemm_bench.py 2>/dev/null; done; } > $O/${TAG}_gemm_variants.txt
{ echo "# tools/probes/mfma_f64_peak.hip"; $R/variants/mfma_peak; echo "# tools/probes/mfma_f64_feed.hip"; $R/variants/mfma_feed; echo "# tools/gpu_two_chain_probe.py"; timeout 300 python $R/tools/gpu_two_chain_probe.py; } > $O/${TAG}_mfma_probes.txt 2>&1
{ echo "# python tools/gpu_inverse_bench.py on 1 x MI355X (${TAG}), through the C-ABI, best of 3 timed repetitions, one chain"; timeout 300 python $R/tools/gpu_inverse_bench.py 2>/dev/null; echo "== DNAGPU_DAG=1 (tile-DAG path, opt-in)"; DNAGPU_DAG=1 timeout 300 python $R/tools/gpu_inverse_bench.py 2>/dev/null; } > $O/${TAG}_inverse_rates.txt
ls -la $O
