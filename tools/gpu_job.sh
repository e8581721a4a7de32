#!/bin/bash
# the whole GPU suite, as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/suite
( time timeout 1700 python -m pytest tests/ -q -m gpu --durations=25 ) > gpurun_out/suite/suite.txt 2>&1
tail -45 gpurun_out/suite/suite.txt
