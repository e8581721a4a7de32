#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/j3
timeout 900 python -m pytest tests/test_gpu_adjust.py -q -m gpu -x -k "lock_step or singular or many_small" 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_adjust.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -q -m gpu -x --deselect tests/test_gpu_fullsize.py::test_cfg4_full_size_properties 2>&1 | tail -8
