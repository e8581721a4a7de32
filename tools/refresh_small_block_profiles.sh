R=$PWD; TAG=r06; O=$R/gpurun_out/profiles_new2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; B="python $R/bench.py"
timeout 600 $B --workload smallblocks --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_smallblocks.json
timeout 900 $B --workload dnasegment150 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150.json
timeout 1500 $B --workload dnasegment150_10x --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150_10x.json
for w in smallblocks dnasegment150 dnasegment150_10x; do
  DNAGPU_PHASE_TIMES=1 timeout 600 $B --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg 2>&1 | grep "^\[phase\]" | tail -40 > $O/${TAG}_${w}_phase_times.txt
done
F="--steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-refactor-leg"
for w in smallblocks dnasegment150; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o p --output-format csv -- timeout 600 $B --workload $w $F > $O/kt_$w.log 2>&1
  python $R/tools/rocprof_summary.py stats /tmp/kt_$w $O/${TAG}_${w}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload $w ${F}   (${TAG}; the trace covers PrepareAdjustment, ONE adjustment and the closing statistics)"
done
ls $O
