#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
DNAGPU_POISON_ALLOC=1 timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_distributed.py tests/test_gpu_exact.py tests/test_gpu_matrix.py tests/test_boundary_cpp.py tests/test_gpu_fullsize.py -q -m gpu -x --deselect tests/test_gpu_fullsize.py::test_cfg4_full_size_properties 2>&1 | tail -4
