#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_adjust.py tests/test_gpu_terrestrial.py -q -m gpu -k "lock_step or singular or many_small" 2>&1 | tail -30
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "dnasegment150" 2>&1 | tail -3
