#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/j13
timeout 1200 python -m pytest tests/test_gpu_matrix.py tests/test_gpu_kernels.py tests/test_gpu_adjust.py tests/test_gpu_batch.py tests/test_gpu_terrestrial.py tests/test_gpu_exact.py -q -m gpu -x 2>&1 | tail -5
for w in dnasegment150 smallblocks cfg3; do
  DNAGPU_PHASE_TIMES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-refactor-leg --no-one-chain > gpurun_out/j13/$w.json 2> gpurun_out/j13/$w.err
  cut -c1-250 gpurun_out/j13/$w.json; grep "phase" gpurun_out/j13/$w.err | tail -24 | grep "iteration 1\|variance\|AdjustNetwork"
done
timeout 300 python tools/gpu_inverse_bench.py 2>/dev/null | tail -12
