#!/bin/bash
# Same-box A/B of causal-attention builds: LLM tests on each candidate, then interleaved kernel timings and the Cfg4 prefill.
cd "${GRAFT_REPO_ROOT:-.}"
for l in "$@"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 300 python -u -m pytest tests/test_gpu_llm.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -1 | sed "s/^/$l tests: /"; done
for r in 1 2; do for l in "$@"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 200 python tools/bench_causal.py 2>/dev/null | sed "s/^/$l /"; done; done
for r in 1 2; do for l in "$@"; do HK_LIB_PATH=$PWD/hawkeye_amd/libhawkeye_hip_${l}_ab.so timeout -k 10 200 python tools/bench_prefill.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l prefill_ms', d['prefill_ms'], 'tflops', d['prefill_tflops'])"; done; done
