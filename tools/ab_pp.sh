#!/bin/bash
# interleaved kernel timings of GEMM library builds (timing probes included: no result checks):  tools/ab_pp.sh <name>...
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for l in "$@"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 200 python tools/duo_check.py pp 2>/dev/null | grep "^\[" | grep -v "proj\|fc2"; done; done
