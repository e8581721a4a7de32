#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
DNAGPU_PHASE_TIMES=1 timeout 600 python -m pytest tests/test_gpu_terrestrial.py -q -m gpu -x -k "lock_step" 2>&1 | grep -v "^\[phase\] [ivRAv]" | tail -30
