#!/bin/bash
# In-situ A/B of environment switches on one box, interleaved: tools/ab_env.sh <rounds> <out-prefix> "VAR=a" "VAR=b" ...
set -u
R=$1; P=$2; shift 2
for i in $(seq 1 $R); do
  k=0
  for E in "$@"; do
    k=$((k+1))
    env $E timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 8 > ${P}_v${k}_$i.json || exit 1
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("${P}_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], {t: v["us"] for t, v in k.items() if t.startswith("vit") and t != "vit.patch_embed"})
PY
