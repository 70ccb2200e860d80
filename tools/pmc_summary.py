#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean of each counter."""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "hk_" not in n:
            continue
        m = re.search(r"(hk_\w+(<[^>]*>)?)", n)
        key = (m.group(1), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(agg.items()):
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
