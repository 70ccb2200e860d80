#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 300 python -u -m pytest tests/test_gpu_train_kernels.py tests/test_gpu_train.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
for f in 0 1 0 1; do HK_PACK_BATCH=$f timeout -k 10 200 python tools/bench_train_step.py --steps 20 2>/dev/null | grep '^{' | sed "s/^/pack_batch=$f /"; done > gpurun_out/r03_cfg5_pack_ab.txt
cut -c1-14,100-160,330-460 gpurun_out/r03_cfg5_pack_ab.txt
