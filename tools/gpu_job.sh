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
cp $O/${TAG}_hbm_traffic.json $R/profiles/      # (so that the bench below quotes it: same sources)
cd $R
timeout 900 python bench.py 2> $O/default_run.err | tail -1 > $O/${TAG}_bench_default_run.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/profiles_new/r05_bench_default_run.json'))
print(r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic'], r['roofline']['traffic_source'])
PY
