#!/bin/bash
# one gpurun job (edit per need)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
( ORACLE_THREADS=16 timeout 2400 python tools/make_fullsize_golden.py cfg3 gpurun_out/cfg3_oracle.npz > gpurun_out/golden_cfg3.log 2>&1 ) &
GOLD=$!
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_matrix.py -q -x --durations=10 > gpurun_out/t_dist.log 2>&1
echo "dist rc=$?" > gpurun_out/job.status
timeout 1500 python -m pytest tests -q -m gpu --durations=25 --deselect tests/test_gpu_distributed.py --deselect tests/test_gpu_matrix.py > gpurun_out/t_all.log 2>&1
echo "all rc=$?" >> gpurun_out/job.status
timeout 600 python tools/gpu_chain_phase.py > gpurun_out/chain_phase.log 2>&1
for c in 4 6 8; do
  DNAGPU_CHAINS=$c timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_chains$c.json
done
wait $GOLD
tail -n 4 gpurun_out/t_dist.log gpurun_out/t_all.log gpurun_out/chain_phase.log gpurun_out/golden_cfg3.log
cat gpurun_out/job.status
