#!/bin/bash
# cfg4 (BASELINE.json configs[3]) / cfg5 (configs[4]: + --variance-propagation) at full size on ONE GPU, staged: the packed variance
# matrices (373 GB) go to page-locked host memory as far as the container's memory limit allows and stay packed in HBM past it.
# usage (on the GPU box, from the repo root): TAG=cfg4 bash tools/run_cfg4_1gpu.sh [extra bench.py flags]
# A guard ends the run before the container's memory limit does (a container killed for memory takes the whole box with it).
set -u
out=gpurun_out/cfg4
mkdir -p $out
tag=${TAG:-cfg4}
export DNAGPU_PHASE_TIMES=1
{ free -g; nproc; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "memory.max $(cat /sys/fs/cgroup/memory.max 2>&1)"; } > $out/host_before.txt
limit=$(cat /sys/fs/cgroup/memory.max 2>/dev/null)
case "$limit" in ''|max) limit=0;; esac
timeout ${LIMIT:-2400} python bench.py --workload cfg4 --gpus 1 --stage --steps 1 --warmup 0 --no-one-chain --no-refactor-leg "$@" > $out/$tag.json 2> $out/$tag.err &
pid=$!
peak=0
while kill -0 $pid 2>/dev/null; do
    cur=$(cat /sys/fs/cgroup/memory.current 2>/dev/null || echo 0)
    [ "$cur" -gt "$peak" ] && peak=$cur
    if [ "$limit" -gt 0 ] && [ "$cur" -gt $((limit - 20000000000)) ]; then
        echo "memory guard: $cur bytes in use of $limit -- ending the run" >> $out/$tag.err
        pkill -9 -P $pid
        kill -9 $pid
        break
    fi
    sleep 0.5
done
wait $pid
echo "exit $? peak_memory_current_bytes $peak" >> $out/$tag.err
free -g > $out/host_after.txt
tail -40 $out/$tag.err
cat $out/$tag.json
