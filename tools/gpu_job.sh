#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16
echo "== phase times, four chains"
DNAGPU_PHASE_TIMES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | grep -E "phase|ms_per_step" | cut -c1-300 | tail -24
echo "== phase times, one chain"
DNAGPU_MULTI_THREAD=0 DNAGPU_PHASE_TIMES=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-one-chain 2>&1 | grep -E "phase" | tail -8
echo "== bench --gpus 2 on one shared GPU (one process)"
DNAGPU_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_share2.json 2> gpurun_out/bench_share2.err; echo "rc=$?"; tail -3 gpurun_out/bench_share2.err; cut -c1-1500 gpurun_out/bench_share2.json
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
cp gpurun_out/parity_*.json gpurun_out/ 2>/dev/null
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
