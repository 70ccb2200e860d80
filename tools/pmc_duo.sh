#!/bin/bash
# Evidence passes for the two-workgroups-per-CU GEMM against the 256x256 kernel (tools/pmc_duo.py): a kernel trace and
# three counter passes, one rocprofv3 run each.  -> gpurun_out/pmc_duo/summary.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_duo
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 tools/pmc_duo.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
pass() {
  local name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/pmc_duo.py > $OUT/$name.log 2>&1 || { echo "pass $name failed rc=$?"; tail -5 $OUT/$name.log; return 1; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE &&
pass b TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE &&
pass c FETCH_SIZE GRBM_GUI_ACTIVE &&
{ echo "== kernel stats (rocprofv3 --kernel-trace --stats) =="; grep -E "Name|hk_gemm" $OUT/kernel_stats.csv; echo "== counters, mean per launch =="; python3 tools/pmc_summary.py $OUT; } > $OUT/summary.txt && cat $OUT/summary.txt
