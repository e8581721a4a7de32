#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export GPU_MAX_HW_QUEUES=16
mkdir -p gpurun_out/flaky
fails=0
for i in $(seq 1 8); do
  timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_lookahead.py tests/test_gpu_adjust.py tests/test_gpu_distributed.py -q -p no:cacheprovider --tb=short -x > gpurun_out/flaky/seq_$i.log 2>&1
  if grep -q "failed" gpurun_out/flaky/seq_$i.log; then fails=$((fails+1)); echo "run $i FAILED"; grep -E "^FAILED|^E  " gpurun_out/flaky/seq_$i.log | head -12 | cut -c1-900; else echo "run $i ok: $(grep -E 'passed' gpurun_out/flaky/seq_$i.log | tail -1)"; rm gpurun_out/flaky/seq_$i.log; fi
done
echo "failures: $fails of 8"
