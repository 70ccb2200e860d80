#!/bin/bash
# rocprofv3 kernel trace of the decode bench (Vicuna-7B, B=1, prompt 352, 40 new tokens): per-kernel summary -> gpurun_out/<tag>_decode_summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r03}
B=${2:-1}
OUT=gpurun_out/prof_decode_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 tools/bench_decode.py $B 352 40 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $T 1 > gpurun_out/${TAG}_decode_summary.txt
head -24 gpurun_out/${TAG}_decode_summary.txt
