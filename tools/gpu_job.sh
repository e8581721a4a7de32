#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j34
TAG=cfg5 bash tools/run_cfg4_1gpu.sh --variance-propagation > gpurun_out/j34/cfg5_run.log 2>&1
cp gpurun_out/cfg4/cfg5.* gpurun_out/j34/ 2>/dev/null
python - <<'PY'
import json
r=json.load(open('gpurun_out/j34/cfg5.json'))
print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline'].get('frac_min_work'), r['config'].get('variance_propagation_in_step'), r['check'])
PY
grep "phase" gpurun_out/j34/cfg5.err | tail -6
