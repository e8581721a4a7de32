#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j21
( time timeout 1500 python bench.py --workload dnasegment150_10x --steps 3 --warmup 1 > gpurun_out/j21/d10x_full.json 2> gpurun_out/j21/d10x_full.err ) 2>&1 | tail -3
python - <<'PY'
import json
r=json.load(open('gpurun_out/j21/d10x_full.json'))
print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline'].get('frac_one_chain'))
print(json.dumps(r['cpu_baseline'])[:900])
print(r.get('without_factor_reuse'))
PY
tail -3 gpurun_out/j21/d10x_full.err
