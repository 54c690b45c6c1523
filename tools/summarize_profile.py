#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into one text summary (per-kernel
average duration from --kernel-trace --stats, per-kernel PMC averages)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
print(f"# rocprofv3 summary, tag={tag}")


def short(name):
    name = name.split("(")[0].replace("void ", "").replace("vr::", "").replace("unsigned short", "u16").replace("unsigned char", "u8")
    return name[:70]


for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print(f"\n## kernel stats ({os.path.relpath(f, out)})")
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        print(f"{short(r['Name']):72s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):12.0f} "
              f"min_ns={r['MinNs']:>10s} max_ns={r['MaxNs']:>10s} pct={r['Percentage']}")

for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"\n## PMC {os.path.basename(d)}")
        for k, cs in acc.items():
            for c, v in cs.items():
                print(f"{k:72s} {c:28s} n={len(v):4d} avg={sum(v)/len(v):16.1f} min={min(v):16.1f} max={max(v):16.1f}")
