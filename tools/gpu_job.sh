#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
s=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_like.out 2> gpurun_out/driver_like.err
echo "rc=$? wall=$(( $(date +%s) - s )) s"
tail -c 600 gpurun_out/driver_like.err
tail -1 gpurun_out/driver_like.out | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('traffic'), j['cpu_baseline']['value'], j['cpu_baseline'].get('seconds_sample'))"
