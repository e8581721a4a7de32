#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/profiles_new
mkdir -p $O
T0=$SECONDS
timeout 1500 python -m pytest tests -q -m gpu -x --durations=8 2>&1 | tail -20 > gpurun_out/gpu_suite.txt
echo "suite: $((SECONDS - T0)) s" >> gpurun_out/gpu_suite.txt
tail -6 gpurun_out/gpu_suite.txt
cd /tmp && export TMPDIR=/tmp
TAG=r05
B="python $R/bench.py"
F="--steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-refactor-leg"
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-refactor-leg"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- timeout 600 $B $F > $O/pmc_$c.log 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python $R/tools/rocprof_summary.py pmc /tmp/pmc_$c $O/${TAG}_cfg3_pmc_$lc.txt "rocprofv3 --pmc $c --kernel-trace -- $CMD   (${TAG}, cfg3)"
done
(cd $O && python $R/tools/pmc_traffic_json.py ${TAG}_cfg3_pmc_fetch_size.txt ${TAG}_cfg3_pmc_write_size.txt ${TAG}_hbm_traffic.json cfg3 > /dev/null)
tail -3 $O/${TAG}_hbm_traffic.json
for w in smallblocks dnasegment150; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o p --output-format csv -- timeout 600 $B --workload $w $F > $O/kt_$w.log 2>&1
  python $R/tools/rocprof_summary.py stats /tmp/kt_$w $O/${TAG}_${w}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w ${F}   (${TAG}; the trace covers PrepareAdjustment, ONE adjustment and the closing statistics)"
done
cd $R
timeout 900 python bench.py 2> $O/default_run.err | tail -1 > $O/${TAG}_bench_default_run.json
timeout 600 python bench.py --workload smallblocks --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_smallblocks.json
timeout 900 python bench.py --workload dnasegment150 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150.json
DNAGPU_PHASE_TIMES=1 timeout 900 python bench.py --workload dnasegment150_10x --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain > $O/${TAG}_bench_dnasegment150_10x.json 2> $O/d10x.err
grep "phase\|bench\]" $O/d10x.err | tail -26 > $O/${TAG}_dnasegment150_10x_phase_times.txt
for w in dnasegment150 smallblocks; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg 2>&1 | grep "^\[phase\]" | tail -40 > $O/${TAG}_${w}_phase_times.txt
done
for f in $O/${TAG}_bench_*.json; do echo $f; cut -c100-230 $f; echo; done
echo "total $((SECONDS - T0)) s"
