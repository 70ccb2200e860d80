#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_tiny
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tr -- python3 tools/tiny_gemm.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
T=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$T" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("hk_gemm") or "hk_gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# groups of 100 consecutive launches
for g in range(0, len(rows), 100):
    grp = rows[g:g + 100]
    if len(grp) < 100: break
    dur = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
    gaps = sorted((int(grp[i + 1]["Start_Timestamp"]) - int(grp[i]["End_Timestamp"])) / 1e3 for i in range(99))
    print(f"group {g // 100}: {grp[0]['Kernel_Name'][:48]} wg={grp[0].get('Grid_Size_X', grp[0].get('Grid_Size'))}/{grp[0].get('Workgroup_Size_X', grp[0].get('Workgroup_Size'))}: duration med {dur[50]:.1f} us min {dur[0]:.1f}; gap med {gaps[49]:.1f} us")
PY
