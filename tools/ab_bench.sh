#!/bin/bash
# Same-box, interleaved end-to-end A/B: the session-start tree (_ab_base/, built separately) against the working tree,
# with and without the folded LayerNorm.  usage: tools/ab_bench.sh <rounds> <out-prefix>
set -u
R=${1:-2}; P=${2:-gpurun_out/ab}
for i in $(seq 1 $R); do
  (cd _ab_base && timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 10) > ${P}_base_$i.json || exit 1
  HK_FOLD_LN=0 timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 10 > ${P}_nofold_$i.json || exit 1
  HK_FOLD_LN=1 timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 10 > ${P}_fold_$i.json || exit 1
done
python - <<PY
import json, glob
for f in sorted(glob.glob("${P}_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["roofline"]["kernels"]
    print(f.split("/")[-1], d["value"], d["ms_per_step"], {t: v["us"] for t, v in k.items() if t.startswith("vit") and t != "vit.patch_embed"})
PY
