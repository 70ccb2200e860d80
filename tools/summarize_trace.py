#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid) -> calls, average and total time.
usage: tools/summarize_trace.py <kernel_trace.csv> [steps]   (steps = profiled encode_img calls)"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(hk_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    agg = collections.defaultdict(list)
    for r in rows:
        wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
        agg[(short(r["Kernel_Name"]), wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':44s} {'workgroups':>10s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'ms/step':>8s} {'%':>6s}")
    for (k, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / tot < 5e-4:
            continue
        print(f"{k:44s} {w:10d} {len(v):6d} {sum(v)/len(v):10.1f} {sum(v)/1e3:9.2f} {sum(v)/1e3/steps:8.2f} {100*sum(v)/tot:6.2f}")
    print(f"total kernel time {tot/1e3:.2f} ms over {steps} step(s) = {tot/1e3/steps:.2f} ms/step")


if __name__ == "__main__":
    main()
