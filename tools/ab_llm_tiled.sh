#!/bin/bash
# Same-box A/B of the prefill with row-major vs 64x64-tiled projection weights (HK_LLM_TILED), B = 8 / 2 / 1.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for b in 8 1 2; do for t in 0 1; do
  B=$b HK_LLM_TILED=$t timeout -k 10 300 python tools/bench_prefill.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$b tiled=$t', {k: v for k, v in d.items() if 'ms' in k or 'tflops' in k.lower()})"
done; done; done | tee gpurun_out/ab_llm_tiled.txt
