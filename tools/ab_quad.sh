#!/bin/bash
# Same-box A/B of GEMM library builds (tools/build_variant.sh): output digests of every build (must agree), the GEMM kernel tests on
# the candidates, then interleaved kernel timings and bench steps.   tools/ab_quad.sh <base> <cand>...
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for l in "$@"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 300 python tools/duo_check.py digest 2>/dev/null | grep DIGEST > gpurun_out/digest_$l.txt || exit 1; done
base=$1
for l in "${@:2}"; do if cmp -s gpurun_out/digest_$base.txt gpurun_out/digest_$l.txt; then echo "digest $l == $base ($(wc -l < gpurun_out/digest_$l.txt) outputs)"; else echo "digest $l DIFFERS from $base"; diff gpurun_out/digest_$base.txt gpurun_out/digest_$l.txt; exit 1; fi; done
for l in "${@:2}"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py -m gpu -x -q --timeout 300 --timeout-method thread -k gemm 2>&1 | tail -1 | sed "s/^/$l tests: /"; done
for r in 1 2; do for l in "$@"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 200 python tools/duo_check.py pp 2>/dev/null | grep "^\["; done; done | tee gpurun_out/ab_quad_kernels.txt
for r in 1 2; do for l in "$@"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-insitu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['value'], d['ms_per_step'])"; done; done | tee gpurun_out/ab_quad_bench.txt
