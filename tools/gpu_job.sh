#!/bin/bash
# one gpurun job (edit per need): tests with durations, chain-phase tool, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
timeout 1500 python -m pytest tests -q -m gpu --durations=25 -x > gpurun_out/t_all.log 2>&1
echo "tests rc=$?" > gpurun_out/job.status
timeout 600 python tools/gpu_chain_phase.py > gpurun_out/chain_phase.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err
for t in 160 384 768; do
  DNAGPU_SMALL_TILES=$t DNAGPU_MULTI_THREAD=0 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_one_chain_small$t.json
  DNAGPU_SMALL_TILES=$t timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_four_chains_small$t.json
done
tail -n 30 gpurun_out/t_all.log
cat gpurun_out/job.status
