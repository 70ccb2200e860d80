#!/bin/bash
# Same-box A/B of environment switches on B=1 call latency: tools/ab_env_b1.sh "VAR=a" "VAR=b" ...  (2 interleaved rounds)
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for e in "$@"; do env $e timeout -k 10 120 python tools/b1_bench.py ${SHAPES:-1x8,1x12,1x16} 2>/dev/null | sed "s/^/$e /"; done; done
