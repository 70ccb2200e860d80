#!/bin/bash
# PMC passes over the ViT attention launch (tools/pmc_gemm.py attn); one rocprofv3 run per pass.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_attn
mkdir -p $OUT
pass() {
  local name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 tools/pmc_gemm.py attn > $OUT/$name.log 2>&1 || { echo "pass $name failed rc=$?"; tail -5 $OUT/$name.log; return 1; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE &&
pass b SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE &&
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt && cat $OUT/summary.txt
