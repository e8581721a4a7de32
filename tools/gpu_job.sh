#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
s=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_like.out 2> gpurun_out/driver_like.err
echo "rc=$? wall=$(( $(date +%s) - s )) s"
tail -1 gpurun_out/driver_like.out | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('traffic'), j['cpu_baseline']['value'], j['config'].get('variance_matrices'))"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
