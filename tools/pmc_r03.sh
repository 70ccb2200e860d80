#!/bin/bash
# PMC passes over tools/pmc_r03.py; one rocprofv3 run per pass (counter limits per block: MI355X_MICROARCH.md).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_r03
rm -rf $OUT; mkdir -p $OUT
pass() {
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/pmc_r03.py > $OUT/$name.log 2>&1 || { echo "pass $name failed rc=$?"; tail -5 $OUT/$name.log; return 1; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
pass b SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE &&
python3 tools/pmc_summary.py $OUT > gpurun_out/r03_pmc.txt && wc -l gpurun_out/r03_pmc.txt
