#!/usr/bin/env python3
"""Per-dispatch rows of a rocprofv3 --pmc CSV for hk_ kernels: duration, counters, GUI_ACTIVE-derived clock."""
import collections, csv, glob, sys
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hk_" not in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = rows.setdefault(k, {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
for k, e in sorted(rows.items()):
    clk = e.get("GRBM_GUI_ACTIVE", 0) / e["us"] / 1e3
    print(k, " ".join(f"{n}={v:.4g}" for n, v in e.items()), f"clk_GHz(GUI/time)={clk:.3f}")
