#!/usr/bin/env python3
"""Timeline of ONE small encode_img call from a rocprofv3 --kernel-trace CSV: for the last call in the trace, every
kernel's duration and the gap to its successor on the device, i.e. how much of a B = 1 call is kernels and how much
is the dispatch-to-dispatch distance of ~300 dependent launches.
usage: tools/call_timeline.py <kernel_trace.csv> <launches per call>"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(hk_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:50]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
rows = rows[-n:]
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for a, b in zip(rows, rows[1:] + [None]):
    wgs = int(a["Grid_Size_X"]) // max(int(a["Workgroup_Size_X"]), 1)
    k = (short(a["Kernel_Name"]), wgs)
    agg[k][0] += 1
    agg[k][1] += (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    if b is not None:
        agg[k][2] += max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
print(f"{n} launches: span {(t1 - t0) / 1e6:.3f} ms, kernels busy {busy / 1e6:.3f} ms, gaps {(t1 - t0 - busy) / 1e6:.3f} ms "
      f"({(t1 - t0 - busy) / n / 1e3:.2f} us per launch)")
print(f"{'kernel':44s} {'wgs':>7s} {'n':>4s} {'avg_us':>8s} {'gap_after_us':>12s} {'total_ms':>9s}")
for (k, w), (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k:44s} {w:7d} {c:4d} {d / c:8.1f} {g / c:12.1f} {(d + g) / 1e3:9.3f}")
