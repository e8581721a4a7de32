#!/bin/bash
# round-5 job 4
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/j4
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_terrestrial.py -q -m gpu -x --durations=5 2>&1 | tail -15 > gpurun_out/j4/tests.txt
cat gpurun_out/j4/tests.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "dnasegment150 or smallblocks or cfg3q" 2>&1 | tail -5 > gpurun_out/j4/fullsize.txt
cat gpurun_out/j4/fullsize.txt
for w in dnasegment150 smallblocks; do
DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg > gpurun_out/j4/$w.phases.json 2> gpurun_out/j4/$w.err
grep "phase" gpurun_out/j4/$w.err | tail -14
timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-one-chain --no-refactor-leg > gpurun_out/j4/$w.json 2>> gpurun_out/j4/$w.err
python -c "
import json;d=json.load(open('gpurun_out/j4/$w.json'));print('$w',d['value'],d['ms_per_step'],d['roofline']['frac'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload dnasegment150 --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-refactor-leg > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/j4/prof.err
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py stats /tmp/kt gpurun_out/j4/ds150_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python bench.py --workload dnasegment150 --steps 1 --warmup 0 --no-cpu-baseline --no-one-chain --no-refactor-leg (r05)"; head -16 gpurun_out/j4/ds150_kernel_stats.txt | cut -c1-200
