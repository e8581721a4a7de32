#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "default_cut" --durations=2 2>&1 | tail -6
cat gpurun_out/default_cut_project_full_size.json
