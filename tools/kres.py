#!/usr/bin/env python3
"""Per-kernel resource usage and static instruction mix from the device assembly (make -C ppqsflhe_amd/csrc asm).

usage: tools/kres.py [engine.s] [name-filter ...]
Columns: VGPRs, spilled VGPRs, scratch bytes/lane, LDS bytes, waves/SIMD (512 VGPRs per lane and SIMD, granule 8), and the
STATIC count of instructions in the kernel body by issue class (v_mad_u64_u32 and f64 ops listed on their own) -- loops
count once, so compare kernels with themselves across edits, not with the dynamic counters of tools/sqsum.py.
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return out.stdout.split("\n")
    except OSError:
        return names


def main():
    path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".s") else "ppqsflhe_amd/csrc/engine.s"
    filters = [a for a in sys.argv[1:] if not a.endswith(".s")]
    meta = {}
    body = {}
    cur = None
    label = re.compile(r"^(_Z\w+):")
    with open(path) as f:
        entries = []
        for line in f:
            m = label.match(line)
            if m:
                cur = m.group(1)
                body[cur] = {"valu": 0, "mad64": 0, "f64": 0, "salu": 0, "vmem": 0, "lds": 0, "wait": 0, "scratch": 0}
                continue
            s = line.strip()
            if line.startswith("  - ."):  # a new entry of amdhsa.kernels (keys are sorted: .name comes in the middle)
                entry = {}
                entries.append(entry)
                s = s[2:]
            if entries and cur is None and s.startswith("."):
                k, _, v = s.partition(":")
                v = v.strip()
                if k == ".name":
                    entries[-1][k] = v
                elif k in (".vgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size",
                           ".sgpr_count", ".agpr_count"):
                    entries[-1][k] = int(v)
            if cur is None or not s or s.startswith((".", ";")):
                if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
                    cur = None
                continue
            op = s.split()[0]
            b = body[cur]
            if op.startswith("v_"):
                b["valu"] += 1
                if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
                    b["mad64"] += 1
                if "_f64" in op:
                    b["f64"] += 1
            elif op.startswith("s_waitcnt"):
                b["wait"] += 1
            elif op.startswith("s_"):
                b["salu"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                b["vmem"] += 1
            elif op.startswith("scratch_"):
                b["scratch"] += 1
            elif op.startswith("ds_"):
                b["lds"] += 1
    meta = {e['.name']: e for e in entries if '.name' in e}
    names = [n for n in meta if n in body]
    pretty = demangle(names)
    rows = []
    for n, p in zip(names, pretty):
        p = re.sub(r"^void ", "", p)
        p = re.sub(r"\(.*$", "", p)
        if filters and not any(f in p for f in filters):
            continue
        m, b = meta[n], body[n]
        v = m.get(".vgpr_count", 0) + 0
        gran = (v + 7) // 8 * 8
        waves = min(8, 512 // gran) if gran else 8
        rows.append((p, v, m.get(".vgpr_spill_count", 0), m.get(".private_segment_fixed_size", 0),
                     m.get(".group_segment_fixed_size", 0), waves, b))
    print(f"{'kernel':58s} {'vgpr':>4s} {'spill':>5s} {'scr':>4s} {'lds':>6s} {'w':>2s} | {'valu':>6s} {'mad64':>5s} {'f64':>5s} "
          f"{'salu':>5s} {'vmem':>5s} {'lds':>5s} {'scr':>4s} {'wait':>4s}")
    for p, v, sp, scr, lds, w, b in sorted(rows):
        print(f"{p[:58]:58s} {v:4d} {sp:5d} {scr:4d} {lds:6d} {w:2d} | {b['valu']:6d} {b['mad64']:5d} {b['f64']:5d} {b['salu']:5d} "
              f"{b['vmem']:5d} {b['lds']:5d} {b['scratch']:4d} {b['wait']:4d}")


if __name__ == "__main__":
    main()
