#!/bin/bash
# Interleaved A/B of the attention kernels on one device.  Variant = "HK_ATTN_PIPE HK_ATTN_LOADER HK_ATTN_OCC HK_ATTN_MODE":
# pipe 0 = per-tile kernel (loader 1 = LDS-DMA, 2 = LDS-DMA + resident -m block; mode 1 = overflow-redo softmax,
# 2 = no-maximum pass with fallback), pipe 2 / 8 = software-pipelined kernel with 4 / 8 waves.
# usage: tools/ab_attn_pipe.sh <rounds> <out-file>
set -u
R=${1:-2}; OUT=${2:-gpurun_out/ab_attn_pipe.txt}
: > $OUT
for i in $(seq 1 $R); do
  for v in "0 1 3 1" "0 1 3 2" "0 1 2 2" "2 1 3 1"; do
    set -- $v
    echo -n "pipe=$1 loader=$2 occ=$3 mode=$4 | " >> $OUT
    HK_ATTN_PIPE=$1 HK_ATTN_LOADER=$2 HK_ATTN_OCC=$3 HK_ATTN_MODE=$4 timeout -k 10 120 python -u tools/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids | tr '\n' '|' >> $OUT || exit 1
    echo >> $OUT
  done
done
cat $OUT
