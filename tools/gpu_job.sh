#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
T0=$SECONDS
SKIP_PMC=1 TAG=r05 bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
echo "total: $((SECONDS - T0)) s"
for f in gpurun_out/profiles_new/r05_bench_*.json; do echo "$(basename $f): $(cut -c100-215 $f)"; done
cat gpurun_out/profiles_new/r05_inverse_rates.txt | tail -4
