#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_tile_dag.py tests/test_boundary_cpp.py tests/test_gpu_multi.py "tests/test_gpu_adjust.py::test_bench_distributed_path_over_rccl" -q -m gpu --durations=5 > gpurun_out/t_part.log 2>&1
echo "part rc=$?" > gpurun_out/job.status
tail -n 40 gpurun_out/t_part.log
cat gpurun_out/job.status
