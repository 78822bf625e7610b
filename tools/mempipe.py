#!/usr/bin/env python3
"""Vector-memory pipeline counters per kernel from rocprofv3 --pmc passes (tools/exp_mempipe.sh): TA / TCP (L1) / TCC (L2)
busy and stall cycles, L1 and L2 hit rates, mean L1-miss latency.  usage: tools/mempipe.py <dir> [<dir> ...]"""
import collections, csv, glob, re, sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
gui_n = collections.defaultdict(int)
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            m = re.search(r"(mk::[A-Za-z0-9_]+(?:<[^>(]*>)?)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                gui_n[k] += 1
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][(f, r["Counter_Name"])] += 1
            key = (r["Dispatch_Id"], f)
            if key not in seen and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                seen.add(key)
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3


def g(v, name):
    return v.get(name, 0.0)


print(f"{'kernel':44s} {'us':>7s} {'GHz':>5s} {'TA busy':>8s} {'TA<-TC addr':>11s} {'TA<-TC data':>11s} {'L1 hit':>7s} {'L1miss lat':>10s} "
      f"{'TCP pend':>9s} {'L2 hit':>7s} {'L2 busy':>8s} {'L2 tagstall':>11s} {'L2 req/clk/XCD':>14s}")
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    passes = len({f for (f, c) in calls[k] if c == "GRBM_GUI_ACTIVE"})  # the clock counter rides in every pass: average
    if not passes:
        continue
    gui = g(v, "GRBM_GUI_ACTIVE") / 8.0 / passes  # cycles per XCD
    if gui <= 0:
        continue
    t = dur[k] / passes
    cu_cycles = gui * 256  # TA / TCP counters are summed over CUs
    acc, miss = g(v, "TCP_TOTAL_CACHE_ACCESSES_sum"), g(v, "TCP_TCC_READ_REQ_sum")
    lat = g(v, "TCP_TCC_READ_REQ_LATENCY_sum") / miss if miss else 0
    req, hit, ms = g(v, "TCC_REQ_sum"), g(v, "TCC_HIT_sum"), g(v, "TCC_MISS_sum")
    print(f"{k:44s} {t:7.0f} {gui / (t * 1e3) if t else 0:5.2f} {g(v, 'TA_TA_BUSY_sum') / cu_cycles:8.2f} "
          f"{g(v, 'TA_ADDR_STALLED_BY_TC_CYCLES_sum') / cu_cycles:11.2f} {g(v, 'TA_DATA_STALLED_BY_TC_CYCLES_sum') / cu_cycles:11.2f} "
          f"{1 - miss / acc if acc else 0:7.2f} {lat:10.0f} {g(v, 'TCP_PENDING_STALL_CYCLES_sum') / cu_cycles:9.2f} "
          f"{hit / (hit + ms) if hit + ms else 0:7.2f} {g(v, 'TCC_BUSY_avr') / gui if gui else 0:8.2f} "
          f"{g(v, 'TCC_TAG_STALL_sum') / (gui * 128):11.2f} {req / gui / 8 if gui else 0:14.1f}")
