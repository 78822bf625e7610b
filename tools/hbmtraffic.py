#!/usr/bin/env python3
"""Per-kernel HBM traffic of one bench step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).

usage: tools/hbmtraffic.py <fetch_dir> <write_dir> <units_per_step> [json_key [profile_tag]] > profiles/rNN_pmc_hbm.txt
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 for every kernel: gfx950 FETCH_SIZE reports half of the bytes read, for the
16-byte coalesced shape (MI355X_MICROARCH.md, HBM section) and -- calibrated on a known byte count,
profiles/r02_calib_fetch.txt -- for this library's 8-byte row and column shapes alike; WRITE_SIZE is exact;
Infinity-Cache hits are counted.  With json_key the per-step total is also written to profiles/hbm_traffic.json (read by
bench.py for roofline.traffic, labelled there as a builder-run profile)."""
import csv, glob, json, os, re, sys


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        m = re.search(r"(mk::[A-Za-z0-9_]+(?:<[^>(]*>)?)", r["Kernel_Name"])
        if m:
            agg[m.group(1)] = agg.get(m.group(1), 0.0) + float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
units = int(sys.argv[3])
rows = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))
print(f"# per-kernel FETCH_SIZE / WRITE_SIZE (KiB) for one step ({units} ciphertexts), bench.py --steps 1 --warmup 0 --no-cpu")
tf = tw = 0.0
for k in rows:
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    tf += f
    tw += w
    print(f"{k:44s} FETCH_SIZE={f:14.1f} WRITE_SIZE={w:14.1f} hbm_MiB_per_ct={(2 * f + w) / 1024 / units:8.2f}")
total = (2 * tf + tw) * 1024
print(f"# total: FETCH_SIZE={tf:.1f} KiB WRITE_SIZE={tw:.1f} KiB -> (2*F+W)*1024 = {total:.0f} B/step = "
      f"{total / units / 2**20:.1f} MiB per ciphertext")
if len(sys.argv) > 4:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = os.path.join(root, "profiles", "hbm_traffic.json")
    rec = json.load(open(p)) if os.path.exists(p) else {}
    rec[sys.argv[4]] = {"hbm_bytes_per_step": total, "units_per_step": units, "fetch_size_kib": tf, "write_size_kib": tw,
                        "profile": "profiles/" + (sys.argv[5] if len(sys.argv) > 5 else "rNN") + "_pmc_hbm.txt",
                        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 "
                                  "--warmup 0 --no-cpu` (mk:: kernels only); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 "
                                  "FETCH_SIZE reports half of the bytes read, calibrated for this library's 8- and 16-byte "
                                  "shapes in profiles/r02_calib_fetch.txt; Infinity-Cache hits are counted"}
    json.dump(rec, open(p, "w"), indent=1)
