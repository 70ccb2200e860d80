set -e
cd "${GRAFT_REPO_ROOT:-.}"
for f in 0 1 0 1; do HK_WGRAD_FUSED=$f timeout -k 10 200 python tools/bench_train_step.py --steps 20 2>/dev/null | grep '^{' | sed "s/^/fused=$f /"; done > gpurun_out/r03_cfg5_ab.txt
tail -4 gpurun_out/r03_cfg5_ab.txt
timeout -k 10 300 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
cat gpurun_out/r03_bench.json | cut -c1-600
