#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['ms_per_step'], j['roofline']['frac'])"; }
run short6144
DNAGPU_PATCH_SHORT_K=0 run short0
DNAGPU_PATCH_SHORT_K=12288 run short12288
DNAGPU_PATCH_SHORT_K=4096 run short4096
DNAGPU_MULTI_THREAD=0 run one_short6144
DNAGPU_MULTI_THREAD=0 DNAGPU_PATCH_SHORT_K=0 run one_short0
DNAGPU_MULTI_THREAD=0 DNAGPU_PATCH_SHORT_K=12288 run one_short12288
python tools/gpu_inverse_bench.py 2>/dev/null | tail -4
