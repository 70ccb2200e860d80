#!/bin/bash
# 192-row tile rule of the persistent GEMM (HK_GEMM_192=0/1): same bits, LLM tests, interleaved Cfg4 prefill timings
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for v in 0 1; do HK_GEMM_192=$v timeout -k 10 300 python tools/duo_check.py digest 2>/dev/null | grep DIGEST > gpurun_out/digest_192_$v.txt || exit 1; done
if cmp -s gpurun_out/digest_192_0.txt gpurun_out/digest_192_1.txt; then echo "digests equal ($(wc -l < gpurun_out/digest_192_1.txt) outputs)"; else echo "digests DIFFER"; diff gpurun_out/digest_192_0.txt gpurun_out/digest_192_1.txt; exit 1; fi
timeout -k 10 300 python -u -m pytest tests/test_gpu_llm.py tests/test_gpu_kernels.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -1
for r in 1 2 3; do for v in 0 1; do HK_GEMM_192=$v timeout -k 10 200 python tools/bench_prefill.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HK_GEMM_192=$v prefill_ms', d['prefill_ms'], 'tflops', d['prefill_tflops'], 'encode_ms', d['encode_img_ms'])"; done; done
