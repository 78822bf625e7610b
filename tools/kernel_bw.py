#!/usr/bin/env python3
"""Per-kernel achieved HBM bandwidth of the hot path: bytes of one step (the two --pmc passes, see hbmtraffic.py for the
(2*FETCH_SIZE + WRITE_SIZE) KiB rule) divided by the kernel's time per step (rocprofv3 --kernel-trace of the same command).

usage: tools/kernel_bw.py <fetch_dir> <write_dir> <trace_dir> <steps_in_trace> > profiles/rNN_kernel_bw.txt"""
import csv, glob, re, sys


def name_of(full):
    m = re.search(r"(mk::[A-Za-z0-9_]+(?:<[^>(]*>)?)", full)
    return m.group(1) if m else None


def counters(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        k = name_of(r["Kernel_Name"])
        if k:
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    return agg


fetch, write = counters(sys.argv[1]), counters(sys.argv[2])
steps = int(sys.argv[4])
dur, calls = {}, {}
f = glob.glob(sys.argv[3] + "/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = name_of(r["Kernel_Name"])
    if k:
        dur[k] = dur.get(k, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        calls[k] = calls.get(k, 0) + 1
print("# per kernel: HBM bytes of one step = (2*FETCH_SIZE + WRITE_SIZE) KiB (separate --pmc passes), time per step from the")
print("# kernel trace of the same command, achieved = bytes / time; HBM peak 8 TB/s (6.29 TB/s measured copy ceiling)")
print(f"{'kernel':46s} {'calls/step':>10s} {'ms/step':>9s} {'read MiB':>10s} {'write MiB':>10s} {'TB/s':>6s}")
tb = tt = 0.0
for k in sorted(dur, key=lambda k: -dur[k]):
    if k not in fetch and k not in write:
        continue
    rd, wr = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    t = dur[k] / steps
    tb += rd + wr
    tt += t
    print(f"{k:46s} {calls[k] / steps:10.1f} {t * 1e3:9.3f} {rd / 2**20:10.1f} {wr / 2**20:10.1f} {(rd + wr) / t / 1e12:6.2f}")
print(f"# all mk:: kernels: {tb / 1e9:.2f} GB in {tt * 1e3:.3f} ms of kernel time per step = {tb / tt / 1e12:.2f} TB/s")
