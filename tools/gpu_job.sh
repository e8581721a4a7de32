#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=$R/gpurun_out/profiles_new
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_adjust.py tests/test_gpu_batch.py tests/test_gpu_kernels.py tests/test_gpu_terrestrial.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "record" 2>&1 | tail -3
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
cd $R
for w in dnasegment150 smallblocks; do timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain 2>/dev/null | cut -c100-240; done
