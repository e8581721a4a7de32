#!/bin/bash
# round-5 job 7: the small blocks of iterations >= 2 in one launch
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/j7
timeout 900 python -m pytest tests/test_gpu_adjust.py -q -m gpu -x -k "many_small_blocks or factor_reuse or random_segmentations" --durations=5 2>&1 | tail -30 > gpurun_out/j7/tests.txt
cat gpurun_out/j7/tests.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "dnasegment150 or smallblocks" --durations=4 2>&1 | tail -12
for w in dnasegment150 smallblocks; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/j7/$w.json 2> gpurun_out/j7/$w.err
  cut -c1-1500 gpurun_out/j7/$w.json; grep -i "phase\|iteration" gpurun_out/j7/$w.err | tail -12
done
