#!/bin/bash
# Per-kernel register / scratch / LDS figures of one HIP source as the build compiles it:  tools/kernel_regs.sh hk_gemm.hip [extra flags]
set -e
src=$1; shift
cd "$(dirname "$0")/../hawkeye_amd/csrc"
tmp=$(mktemp -d)
extra=""
case $src in hk_gemm.hip) extra="-fno-honor-nans -fno-slp-vectorize";; hk_attn*.hip) extra="-fno-honor-nans";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra "$@" -save-temps=obj -c $src -o $tmp/out.o
awk '/^ +\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.vgpr_spill_count:/ {printf "%-90s vgpr %3d sgpr %3d scratch %5d spill %d\n", name, v, s, p, $2}' $tmp/*gfx950.s | sed 's/_ZN12_GLOBAL__N_1//'
rm -rf $tmp
