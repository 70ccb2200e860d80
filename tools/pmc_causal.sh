#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for l in "$@"; do
  OUT=gpurun_out/pmc_causal_$l; rm -rf $OUT; mkdir -p $OUT
  HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -o p -- python3 tools/bench_causal.py > $OUT/b.log 2>&1 || { echo "pmc failed"; tail -3 $OUT/b.log; exit 1; }
  python3 tools/pmc_summary.py $OUT | sed "s/^/$l /" | grep -A4 causal | head -12
done
