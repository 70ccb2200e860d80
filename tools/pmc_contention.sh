#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_contention; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o p -- python3 tools/pmc_contention.py > $OUT/a.log 2>&1 || { echo failed; tail -3 $OUT/a.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "hk_gemm" not in r["Kernel_Name"]: continue
    k = int(r["Dispatch_Id"])
    rows.setdefault(k, {"name": r["Kernel_Name"][:40]})[r["Counter_Name"]] = float(r["Counter_Value"])
    rows[k]["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if "End_Timestamp" in r else 0
names = ["normal", "A-broadcast", "zeros", "zeros+A-broadcast"]
for i, (k, c) in enumerate(rows.items()):
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"{names[(i // 4) % 4]:18s} launch {i % 4}: {c['dur']:7.1f} us  cycles/XCD {cyc/1e6:6.3f} M  clock {cyc / c['dur'] / 1e3 if c['dur'] else 0:5.2f} GHz  MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) * 100:5.1f} %  wait_any/wave {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'] * 100:5.1f} %")
PY
