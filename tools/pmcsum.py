#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter over the mk:: kernels of one run (per kernel name and total)."""
import csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(mk::[A-Za-z0-9_]+(?:<[^>(]*>)?)", r["Kernel_Name"])
    if not m:
        continue
    key = (m.group(1), r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
tot = {}
for (k, c), (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} {c:14s} calls={n:5d} sum={v:16.1f}")
    tot[c] = tot.get(c, 0) + v
for c, v in tot.items():
    print(f"TOTAL {c} = {v:.1f}")
