#!/bin/bash
# Round profile of the headline bench (B=32, T=16): rocprofv3 kernel trace + the two HBM-traffic PMC passes, each a run
# of its own (MI355X_MICROARCH.md, rocprofv3 PMC slots).  usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r01g}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log > gpurun_out/${TAG}_bench_under_trace.json
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $T 8 > gpurun_out/${TAG}_summary.txt   # 1 warm-up + 4 timed + 3 in-situ profile calls
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-insitu > $OUT/$C.log 2>&1 || { echo "pmc $C failed"; tail -5 $OUT/$C.log; exit 1; }
done
python3 tools/traffic_summary.py $OUT/FETCH_SIZE $OUT/WRITE_SIZE 2 "B=32 T=16 Lt=32" gpurun_out/${TAG}_hbm_traffic.txt gpurun_out/${TAG}_traffic.json
head -12 gpurun_out/${TAG}_summary.txt; tail -3 gpurun_out/${TAG}_hbm_traffic.txt
