#!/bin/bash
# Round-3 measurement set on one box: headline profile (kernel trace + HBM-traffic PMC passes), sweeps, cfg 3 / 4.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1 || { echo "profile_round failed"; tail -5 gpurun_out/r03_profile_round.log; exit 1; }
echo "profile_round done"
timeout -k 10 300 python tools/sweep.py > gpurun_out/r03_sweep.json 2> gpurun_out/r03_sweep.err || { echo "sweep failed"; tail -5 gpurun_out/r03_sweep.err; exit 1; }
echo "sweep done"
for s in 1 0 1 0; do HK_ATTN_SPLIT=$s timeout -k 10 120 python tools/b1_bench.py 2>/dev/null | sed "s/^/HK_ATTN_SPLIT=$s /"; done > gpurun_out/r03_b1_keysplit_ab.txt || { echo "b1 failed"; exit 1; }
echo "b1 done"
timeout -k 10 300 python tools/bench_prefill.py > gpurun_out/r03_cfg4_prefill.json 2> gpurun_out/r03_cfg4.err || { echo "prefill failed"; tail -5 gpurun_out/r03_cfg4.err; exit 1; }
echo "prefill done"
timeout -k 10 300 python tools/bench_grounding.py > gpurun_out/r03_cfg3_grounding.json 2> gpurun_out/r03_cfg3.err || { echo "grounding failed"; tail -5 gpurun_out/r03_cfg3.err; exit 1; }
echo "grounding done"
