#!/bin/bash
# one gpurun job (edit per need)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 600 python -m pytest tests/test_gpu_matrix.py tests/test_gpu_kernels.py -q -x > gpurun_out/t_fused.log 2>&1
echo "fused rc=$?" > gpurun_out/job.status
if grep -q "fused rc=0" gpurun_out/job.status; then
  timeout 1500 python -m pytest tests -q -m gpu --durations=12 --deselect tests/test_gpu_matrix.py --deselect tests/test_gpu_kernels.py > gpurun_out/t_all.log 2>&1
  echo "all rc=$?" >> gpurun_out/job.status
  for f in 1 0; do
    { echo "# DNAGPU_FUSE=$f python tools/gpu_inverse_bench.py (one chain, best of 3)"; DNAGPU_FUSE=$f timeout 300 python tools/gpu_inverse_bench.py 2>/dev/null; } > gpurun_out/inverse_rates_fuse$f.txt
    DNAGPU_FUSE=$f DNAGPU_MULTI_THREAD=0 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_one_chain_fuse$f.json
    DNAGPU_FUSE=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_four_chains_fuse$f.json
  done
  timeout 600 python tools/gpu_chain_phase.py > gpurun_out/chain_phase.log 2>&1
fi
tail -n 5 gpurun_out/t_fused.log gpurun_out/t_all.log
cat gpurun_out/job.status
