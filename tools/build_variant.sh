#!/bin/bash
# tools/build_variant.sh <suffix> <extra hipcc flags...>: experimental build of the library as gpurun_out/libdnagpu_<suffix>.so
# (kernel tuning experiments; run with DNAGPU_LIB_OVERRIDE=<path> python tools/gpu_gemm_bench.py)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
S=$1; shift
mkdir -p $R/variants
SRCS=""
for f in $R/dynadjust_amd/csrc/*.hip $R/dynadjust_amd/csrc/host/*.cpp; do SRCS="$SRCS -x hip $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -I$R/include -I$R/dynadjust_amd/csrc "$@" -o $R/variants/libdnagpu_$S.so $SRCS -lpthread
echo built $R/variants/libdnagpu_$S.so
