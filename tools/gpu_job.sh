#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
for sp in 0.12 0.2 0.25 0.33 0.5; do
DNAGPU_SCHUR_SPLIT=$sp python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('split $sp', j['ms_per_step'], j['roofline']['frac'], j['cholesky_tflops'])"
done
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_gpu_distributed.py -q 2>&1 | grep -E "passed|failed" | tail -1; done
