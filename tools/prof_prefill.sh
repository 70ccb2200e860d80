#!/bin/bash
# rocprofv3 kernel trace of the Cfg4 bench (encode_img B=8 T=16 -> Vicuna-7B prefill S=352): per-kernel summary -> gpurun_out/<tag>_cfg4_summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_cfg4_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 tools/bench_prefill.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $T 1 > gpurun_out/${TAG}_cfg4_summary.txt
head -30 gpurun_out/${TAG}_cfg4_summary.txt
