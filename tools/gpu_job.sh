#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_new
mkdir -p $O
export GPU_MAX_HW_QUEUES=16
TAG=r03
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -3
cd /tmp && export TMPDIR=/tmp
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- timeout 600 python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain > $O/pmc_$c.log 2>&1
  lc=$(echo $c | tr A-Z a-z)
  python $R/tools/rocprof_summary.py pmc /tmp/pmc_$c $O/${TAG}_cfg3_pmc_$lc.txt "rocprofv3 --pmc $c --kernel-trace -- $CMD   (${TAG}, cfg3)"
done
(cd $O && python $R/tools/pmc_traffic_json.py ${TAG}_cfg3_pmc_fetch_size.txt ${TAG}_cfg3_pmc_write_size.txt ${TAG}_hbm_traffic.json cfg3 > /dev/null)
cp $O/${TAG}_hbm_traffic.json $R/profiles/
cd $R
timeout 600 python bench.py --steps 2 --warmup 1 2> $O/bench_cfg3.err | tail -1 > $O/${TAG}_bench_cfg3.json
cut -c1-220 $O/${TAG}_bench_cfg3.json; cat $O/${TAG}_hbm_traffic.json | tail -4
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
