#!/bin/bash
# the job of the moment for `gpurun -- bash tools/gpu_job.sh` (edited per measurement)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/trace
export GPU_MAX_HW_QUEUES=16
{ DNAGPU_DAG_TRACE=gpurun_out/trace/bench timeout 600 python tools/gpu_inverse_bench.py 6656
  python tools/dag_trace_report.py gpurun_out/trace/bench.0003.bin gpurun_out/trace/bench.0007.bin
  rm -f gpurun_out/trace/*.bin
} > gpurun_out/dag_trace.log 2>&1
rm -f gpurun_out/trace/*.bin
tail -n 90 gpurun_out/dag_trace.log
