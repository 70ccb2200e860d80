#!/bin/bash
# Build a variant of the library with extra flags on ONE source:  tools/build_variant.sh <name> <hk_x.hip> [-DFLAG=..]...
#   -> hawkeye_amd/libhawkeye_hip_<name>_ab.so   (for same-box A/B through HK_LIB_PATH; never loaded by the product)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../hawkeye_amd/csrc"
make -s -j8 >/dev/null
extra=""
case $src in hk_gemm.hip) extra="-fno-honor-nans -fno-slp-vectorize";; hk_attn*.hip) extra="-fno-honor-nans";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra "$@" -c $src -o /tmp/${src%.hip}_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$" | grep -v "_stamp\|_diag")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhawkeye_hip_${name}_ab.so /tmp/${src%.hip}_$name.o $objs
echo built ../libhawkeye_hip_${name}_ab.so
