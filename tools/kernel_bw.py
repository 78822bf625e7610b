#!/usr/bin/env python3
"""Per-kernel achieved HBM bandwidth of the hot path: bytes of one step (the two --pmc passes, see hbmtraffic.py for the
(2*FETCH_SIZE + WRITE_SIZE) KiB rule) divided by the kernel's time per step (rocprofv3 --kernel-trace of the same command).

The "alg MiB" column is the kernel's own compulsory traffic in the merged n-client flow at the bench shape (every operand
tile once, every result once; tiles other workgroups re-read from L2 count once): `alg_limbs` below, in limbs of 8N bytes.
measured / alg > 1 is re-reading (or scratch spills); the step's total against SURVEY.md 8(d)'s 92.67 MB per unit is the
bench line's roofline.

usage: tools/kernel_bw.py <fetch_dir> <write_dir> <trace_dir> <steps_in_trace> > profiles/rNN_kernel_bw.txt"""
import csv, glob, re, sys

# bench shape (bench.py defaults): C clients x B ciphertext indices, L Q limbs (q_0 integer class, the rest fp64 class),
# K P limbs, beta digits of alpha limbs
C, B, L, K, BETA, ALPHA, LOGN = 8, 16, 12, 4, 3, 4, 16
LIMB_MIB = 8 * (1 << LOGN) / 2**20


def alg_limbs(name):
    """(read, written) limbs of one STEP for the kernel `name` (all of its calls), or None if the model has no row."""
    items, polys2, nfp, D = C * B, 2 * B, L - 1, L + K
    # ModUp targets per digit by class: digit 0 owns q_0 + 3 fp limbs, digits 1, 2 own 4 fp limbs each
    fp_t = [nfp - 3, nfp - 4, nfp - 4]
    int_t = [K, K + 1, K + 1]
    m = re.match(r"mk::(\w+)<([^>]*)>", name)
    if not m:
        return None
    k, a = m.group(1), [x.strip() for x in m.group(2).split(",")]
    if k in ("k_ntt_row_r", "k_ntt_col_r") and a[1] == "true":   # inverse passes: c1 (+ the dropped limb of rescale, fp class)
        n = nfp * items + polys2 if a[2] == "1" else items
        return n, n
    if k == "k_ntt_row_r" and a[1] == "false":                      # rescale: row pass of the switched limbs + tail
        t = (nfp - 1) if a[2] == "1" else 1
        return 2 * t * polys2, t * polys2
    if k == "k_switch_col":
        t = (nfp - 1) if a[1] == "1" else 1
        return polys2, t * polys2
    if k in ("k_conv_col", "k_conv_col2"):                          # a = LOG_H, N_IN, AR, DevConv, SRCMODE (col2: two targets per workgroup)
        digits = [0] if a[4] == "2" else [1, 2]
        tg = fp_t if a[2] == "1" else int_t
        return sum(ALPHA * items for _ in digits), sum(tg[j] * items for j in digits)
    if k == "k_row3_inner_int":                                     # a = NPARTS, LOGC, INVP, AR
        if a[2] == "true":
            return BETA * K * items + BETA * 2 * K * C, 2 * K * items
        return (BETA - 1) * items + items + BETA * 2 * C, 2 * items
    if k == "k_icol_sum":
        return 2 * K * items, K * polys2
    if k == "k_conv_col_psum2":                                     # fp64-class targets, two per workgroup
        return K * polys2, nfp * polys2
    if k == "k_conv_col_psum":
        return K * polys2, (nfp if a[2] == "1" else 1) * polys2
    if k == "k_conv_col_sum2" or k == "k_conv_col_sum":            # round 2's per-client summed conversion
        fp = k == "k_conv_col_sum2" or a[2] == "1"
        return 2 * K * items, (nfp if fp else 1) * polys2
    if k == "k_row3_tail_once":
        return polys2 + 2 * items + items, polys2
    if k == "k_qsum3_fp" or k == "k_qsum_fp":
        return nfp * ((BETA - 1) * items + 2 * items + BETA * 2 * C + polys2), nfp * polys2
    return None


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
print(f"{'kernel':46s} {'calls/step':>10s} {'ms/step':>9s} {'read MiB':>10s} {'write MiB':>10s} {'TB/s':>6s} {'alg MiB':>9s} "
      f"{'meas/alg':>8s} {'alg TB/s':>8s}")
tb = tt = ta = 0.0
for k in sorted(dur, key=lambda k: -dur[k]):
    if k not in fetch and k not in write:
        continue
    rd, wr = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    t = dur[k] / steps
    tb += rd + wr
    tt += t
    al = alg_limbs(k)
    if al:
        ab = (al[0] + al[1]) * LIMB_MIB * 2**20
        ta += ab
        extra = f" {ab / 2**20:9.1f} {(rd + wr) / ab:8.2f} {ab / t / 1e12:8.2f}"
    else:
        extra = f" {'-':>9s} {'-':>8s} {'-':>8s}"
    print(f"{k:46s} {calls[k] / steps:10.1f} {t * 1e3:9.3f} {rd / 2**20:10.1f} {wr / 2**20:10.1f} {(rd + wr) / t / 1e12:6.2f}" + extra)
print(f"# all mk:: kernels: {tb / 1e9:.2f} GB in {tt * 1e3:.3f} ms of kernel time per step = {tb / tt / 1e12:.2f} TB/s; "
      f"modelled per-kernel compulsory traffic {ta / 1e9:.2f} GB ({ta / 2**20 / (C * B):.1f} MiB per unit; the path's algorithmic "
      f"88.4 MiB per unit counts every intermediate as free)")
