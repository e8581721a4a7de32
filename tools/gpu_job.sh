#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/profiles_new
mkdir -p $O
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
cp $O/${TAG}_hbm_traffic.json $R/profiles/
cd $R
timeout 900 python bench.py 2> $O/default_run.err | tail -1 > $O/${TAG}_bench_default_run.json
timeout 900 python bench.py --workload dnasegment150 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150.json
timeout 1500 python bench.py --workload dnasegment150_10x --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/${TAG}_bench_dnasegment150_10x.json
for w in dnasegment150 dnasegment150_10x; do
  DNAGPU_PHASE_TIMES=1 timeout 900 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg 2>&1 | grep "^\[phase\]\|\[bench\]" | tail -40 > $O/${TAG}_${w}_phase_times.txt
done
python - <<'PY'
import json
for w in ('default_run','dnasegment150','dnasegment150_10x'):
    r=json.load(open(f'gpurun_out/profiles_new/r05_bench_{w}.json'))
    print(w, r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['traffic'])
PY
