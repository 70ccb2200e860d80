#!/bin/bash
# Same-box A/B of environment switches on the headline bench: tools/ab_env_bench.sh "VAR=a" "VAR=b" ...  (2 interleaved rounds)
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for e in "$@"; do env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-insitu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], d['ms_per_step'])"; done; done
