#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -x --durations=6 > gpurun_out/t_dist.log 2>&1
echo "dist rc=$?" > gpurun_out/job.status
timeout 900 python tools/gpu_intra_block_probe.py > gpurun_out/intra_block.log 2>&1
echo "intra rc=$?" >> gpurun_out/job.status
DNAGPU_FORCE_DISTRIBUTED=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/rccl1.err | tail -1 > gpurun_out/bench_cfg3_rccl_one_rank.json
tail -n 6 gpurun_out/t_dist.log gpurun_out/intra_block.log
cat gpurun_out/job.status
