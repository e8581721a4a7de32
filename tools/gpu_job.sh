#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
rm -rf gpurun_out/profiles_new
TAG=r02 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
echo "refresh rc=$?"
python bench.py --workload cfg4_slice --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/profiles_new/r02_bench_cfg4_slice.json
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --variances-every-iteration 2>/dev/null | tail -1 > gpurun_out/profiles_new/r02_bench_cfg3_variances_every_iteration.json
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --stage 2>/dev/null | tail -1 > gpurun_out/profiles_new/r02_bench_cfg3_staged.json
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --variance-propagation 2>/dev/null | tail -1 > gpurun_out/profiles_new/r02_bench_cfg3_variance_propagation.json
DNAGPU_FORCE_DISTRIBUTED=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/profiles_new/r02_bench_cfg3_rccl_one_rank.json
ls gpurun_out/profiles_new | head -40
