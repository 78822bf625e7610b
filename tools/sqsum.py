#!/usr/bin/env python3
"""Per-kernel SQ summary of one bench step from a rocprofv3 --pmc pass (SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY).  usage: tools/sqsum.py <pmc_dir> > profiles/rNN_sq_counters.txt"""
import collections, csv, glob, re, sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    m = re.search(r"(mk::[A-Za-z0-9_]+(?:<[^>(]*>)?)", r["Kernel_Name"])
    if m:
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
tot = sum(v["SQ_BUSY_CYCLES"] for v in agg.values())
print("# SQ counters per kernel, one step (bench.py --steps 1 --warmup 0 --no-cpu)")
print("# valu_util = SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / 8; wait fractions are of SQ_WAVE_CYCLES")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"]):
    w, b = v["SQ_WAVE_CYCLES"], v["SQ_BUSY_CYCLES"]
    print(f"{k:46s} busy={b:9.3g} ({100 * b / tot:4.1f}%) valu_util={v['SQ_ACTIVE_INST_VALU'] / b / 8:4.2f} "
          f"insts_valu={v['SQ_INSTS_VALU']:9.3g} wait_any={v['SQ_WAIT_ANY'] / w:4.2f} wait_inst_any={v['SQ_WAIT_INST_ANY'] / w:4.2f}")
