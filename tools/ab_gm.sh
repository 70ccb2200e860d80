#!/bin/bash
# Same-box A/B of the persistent GEMM's tile raster group height (HK_GEMM_GM: m-panels per group) on the Cfg4 prefill.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for gm in 4 8 16 2; do
  B=${B:-8} HK_GEMM_GM=$gm timeout -k 10 300 python tools/bench_prefill.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gm=$gm', d['encode_img_ms'], d['prefill_ms'])"
done; done | tee gpurun_out/ab_gm.txt
