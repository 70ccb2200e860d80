#!/bin/bash
# Same-box A/B of epilogue builds (tools/build_variant.sh): kernel tests on the candidate, then interleaved kernel timings and bench steps.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
CAND=${1:-fast2}; shift || true
LIBS="${@:-base fast2 slow2}"
HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${CAND}_ab.so timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q --timeout 300 --timeout-method thread -k gemm 2>&1 | tail -2
for r in 1 2; do for l in $LIBS; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 200 python tools/duo_check.py pp 2>/dev/null | grep "^\["; done; done | tee gpurun_out/ab_epi_kernels.txt
for r in 1 2; do for l in $LIBS; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-insitu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'])"; done; done | tee gpurun_out/ab_epi_bench.txt
