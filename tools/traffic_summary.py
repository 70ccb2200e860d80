#!/usr/bin/env python3
"""HBM-side traffic per encode_img step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; CSV output).
Counter values are KiB; FETCH_SIZE is doubled (gfx950 tallies its 128-byte read requests at 64 bytes:
MI355X_MICROARCH.md, HBM / rocprofv3 section); these are fabric-side bytes, Infinity-Cache hits included.
usage: traffic_summary.py <fetch_dir> <write_dir> <steps> <workload> <out.txt> <out.json>"""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hawkeye_amd.lib import source_digest  # noqa: E402

fdir, wdir, steps, workload, out_txt, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]


def short(name):
    m = re.search(r"(hk_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else "other (torch copies/fills)"


def collect(d, counter):
    agg, calls = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = f'{short(r["Kernel_Name"])} [{int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)} wg]'
            agg[k] += float(r["Counter_Value"]) * 1024.0
            calls[k] += 1
    return agg, calls


fetch, calls = collect(fdir, "FETCH_SIZE")
write, _ = collect(wdir, "WRITE_SIZE")
lines = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 1 = {steps} encode_img steps, {workload}",
         "# units: counter value is KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); fabric-side bytes (Infinity-Cache hits included)",
         f"{'kernel [workgroups]':56s} {'calls/step':>10s} {'fetch GB/step (x2)':>20s} {'write GB/step':>14s}"]
tf = tw = 0.0
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
    f, w = 2 * fetch.get(k, 0.0) / steps, write.get(k, 0.0) / steps
    tf, tw = tf + f, tw + w
    lines.append(f"{k[:56]:56s} {calls[k] // steps:10d} {f / 1e9:20.2f} {w / 1e9:14.2f}")
lines.append(f"{'TOTAL':56s} {'':10s} {tf / 1e9:20.2f} {tw / 1e9:14.2f}")
open(out_txt, "w").write("\n".join(lines) + "\n")
per_kernel = {k: {"fetch_bytes_per_launch": 2 * fetch[k] / max(calls[k], 1), "write_bytes_per_launch": write.get(k, 0.0) / max(calls[k], 1),
                  "launches_per_step": calls[k] // steps} for k in fetch}
json.dump({"bytes_per_step": tf + tw, "fetch_bytes": tf, "write_bytes": tw, "workload": workload, "csrc_digest": source_digest(), "source": out_txt.replace("gpurun_out/", "profiles/"),
           "per_kernel": per_kernel}, open(out_json, "w"))
print("\n".join(lines[-4:]))
