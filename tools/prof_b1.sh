#!/bin/bash
# rocprofv3 kernel trace of B=1 T=12 encode_img calls; timeline of the last call -> gpurun_out/<tag>_timeline_b1_t12.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_b1_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tr -- python3 tools/b1_bench.py 1x12 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
N=$(python3 - "$T" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# launches per call: distance between the last two patchify launches
idx = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
print(idx[-1] - idx[-2])
PY
)
python3 - "$T" "$N" <<'PY' > gpurun_out/${TAG}_timeline_b1_t12.txt
import csv, sys, subprocess
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
idx = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
call = rows[idx[-2]:idx[-1]]
import collections, re
def short(name):
    m = re.search(r"(hk_\w+(<[^>]*>)?)", name)
    if m: return m.group(1)
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]
t0, t1 = int(call[0]["Start_Timestamp"]), int(call[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in call)
print(f"{len(call)} launches: span {(t1-t0)/1e6:.3f} ms, kernels busy {busy/1e6:.3f} ms, gaps {(t1-t0-busy)/1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in call:
    k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{'kernel':48s} {'wgs':>6s} {'n':>4s} {'avg_us':>8s} {'total_ms':>9s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:48s} {k[1]:6d} {v[0]:4d} {v[1]/v[0]:8.1f} {v[1]/1e3:9.3f}")
PY
head -30 gpurun_out/${TAG}_timeline_b1_t12.txt
