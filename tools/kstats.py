#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a short per-kernel table.

usage: tools/kstats.py <rocprof_out_dir> [steps_in_run] > profiles/rNN_<name>.txt
Kernel names are shortened to their function name; torch helper kernels are lumped together.
"""
import csv
import glob
import re
import sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not files:
    sys.exit("no *kernel_stats.csv under " + d)
rows = list(csv.DictReader(open(files[0])))


def short(name):
    m = re.search(r"(mk::[A-Za-z0-9_]+(?:<[^>(]*>)?)", name)
    if m:
        return m.group(1)
    if "at::native" in name or "rocclr" in name:
        return "(torch/runtime helper kernels)"
    return name[:60]


agg = {}
for r in rows:
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"])
    a[2] = min(a[2], float(r["MinNs"]))
    a[3] = max(a[3], float(r["MaxNs"]))
tot = sum(a[1] for a in agg.values())
# per-step wall time from the kernel trace: the mk:: dispatches of the run in start order, cut into `steps` equal groups
# (every step launches the same sequence); wall = first kernel start to last kernel end, busy = union of the kernel
# intervals (kernels of a side stream overlap the main stream's), sum = kernel durations added up
tfiles = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
step_lines = []
if steps and tfiles:
    ev = []
    for r in csv.DictReader(open(tfiles[0])):
        if "mk::" in r["Kernel_Name"] and "k_pack_rowb" not in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    ev.sort()
    if ev and len(ev) % steps == 0:
        per = len(ev) // steps
        for i in range(steps):
            g = ev[i * per:(i + 1) * per]
            wall = max(e for _, e in g) - g[0][0]
            ksum = sum(e - s0 for s0, e in g)
            busy, cur_s, cur_e = 0, g[0][0], g[0][1]
            for s0, e in g[1:]:
                if s0 > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = s0, e
                else:
                    cur_e = max(cur_e, e)
            busy += cur_e - cur_s
            step_lines.append(f"# step {i}: {per} mk:: kernels, wall {wall / 1e6:.3f} ms (first start -> last end), kernel sum "
                              f"{ksum / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, gaps {(wall - busy) / 1e6:.3f} ms, "
                              f"overlap {(ksum - busy) / 1e6:.3f} ms")
    elif ev:
        step_lines.append(f"# {len(ev)} mk:: dispatches do not divide into {steps} equal steps: no per-step wall time")
print(f"# source: {files[0]}")
for l in step_lines:
    print(l)
print(f"# total kernel time {tot / 1e6:.3f} ms" + (f" over {steps} steps = {tot / 1e6 / steps:.3f} ms/step" if steps else ""))
print(f"{'kernel':48s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:48s} {a[0]:7d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / tot:6.2f}")
