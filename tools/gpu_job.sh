#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
pids=""
for i in $(seq 1 $(nproc)); do timeout 300 python -c "
while True: pass" & pids="$pids $!"; done
echo "== condensed schedule (default path, batches, four chains) under CPU load"
timeout 120 python tools/gpu_mt_probe.py 100000 0 1 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 | cut -c1-420
echo "== reference schedule, multi-thread, under CPU load (after the fix)"
timeout 120 python tools/gpu_mt_probe.py 100000 0 0 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -6 | cut -c1-420
for p in $pids; do kill $p 2>/dev/null; done
wait 2>/dev/null
echo done
