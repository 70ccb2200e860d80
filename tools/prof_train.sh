#!/bin/bash
# rocprofv3 kernel trace of the Cfg5 step bench (B=2, T=12): per-kernel summary -> gpurun_out/<tag>_cfg5_summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_cfg5_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 tools/bench_train_step.py --steps 10 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $T 13 > gpurun_out/${TAG}_cfg5_summary.txt
head -40 gpurun_out/${TAG}_cfg5_summary.txt
