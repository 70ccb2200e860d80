#!/bin/bash
# In-situ bisect: bench.py (HK_FOLD_LN=0) with several builds of the library on one box, interleaved.
# usage: tools/ab_bench_libs.sh <rounds> <out-prefix> lib1.so lib2.so ...
set -u
R=$1; P=$2; shift 2
for i in $(seq 1 $R); do
  for L in "$@"; do
    n=$(basename $L .so)
    HK_FOLD_LN=${FOLD:-0} HK_LIB_PATH=$PWD/$L timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 8 > ${P}_${n}_$i.json || exit 1
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("${P}_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], {t: v["us"] for t, v in k.items() if t.startswith("vit") and t != "vit.patch_embed"})
PY
