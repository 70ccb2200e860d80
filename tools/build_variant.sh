#!/bin/bash
# Build a variant of the library with extra flags on hk_gemm.hip:  tools/build_variant.sh <name> [-DFLAG=..]...
#   -> hawkeye_amd/libhawkeye_hip_<name>_ab.so   (for same-box A/B through HK_LIB_PATH; never loaded by the product)
set -e
name=$1; shift
cd "$(dirname "$0")/../hawkeye_amd/csrc"
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-honor-nans -fno-slp-vectorize "$@" -c hk_gemm.hip -o /tmp/hk_gemm_$name.o
objs=$(ls *.o | grep -v "^hk_gemm.o$" | grep -v "_stamp\|_diag")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhawkeye_hip_${name}_ab.so /tmp/hk_gemm_$name.o $objs
echo built ../libhawkeye_hip_${name}_ab.so
