#!/bin/bash
# round-5 job 5: the whole GPU suite on the pruned tree + cfg3 bench + inverse rates
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/j5
timeout 2400 python -m pytest tests -q -m gpu --durations=25 -x 2>&1 | tail -60 > gpurun_out/j5/tests.txt
tail -45 gpurun_out/j5/tests.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/j5/cfg3.json 2> gpurun_out/j5/cfg3.err
python -c "
import json;d=json.load(open('gpurun_out/j5/cfg3.json'));r=d['roofline'];print('cfg3',d['value'],d['ms_per_step'],r['frac'],r['frac_gemm_busy'],r['frac_one_chain'],d.get('without_factor_reuse'));print(d.get('roofline_hbm'))"
timeout 600 python tools/gpu_inverse_bench.py > gpurun_out/j5/inverse_rates.txt 2>&1; head -8 gpurun_out/j5/inverse_rates.txt
