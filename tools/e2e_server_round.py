#!/usr/bin/env python3
"""End-to-end server round through the C++ hosts at the C3 shape (BASELINE configs[1]: N=2^16, L=12, dnum=3, 8 clients x 16
ciphertexts), files in tmpfs -> file in tmpfs.

  genCC (RingDim 65536, depth 10, 50-bit scaling) -> keyGen x 9 -> encryptModelWeights x 8 (MKWS envelopes, 16 ciphertexts
  of 12.6 MB each) -> REkeyGen x 8 (every client into the ninth key's domain, so all 8 x 16 ciphertexts are re-encrypted as
  in bench.py's step) -> serverRound, once per arm:
    pipelined, MKCKKS_IO_THREADS = 1, 2, 4, 8, 12     (host/iopipe.hpp; chunks of 2 ciphertext indices)
    pipelined, 8 threads, MKCKKS_ROUND_CHUNK = 1, 3, 6, 18
    MKCKKS_SYNC_IO=1                                   (read_envelope / decode_ct / mkckks_upload, the r02 path)
Every arm's aggregate file must have the same bytes.  Per arm, over --reps runs: serverRound's own "files to file" time
(first read -> last byte written, buffers / keys / kernels resident: what a server process pays per round) as
min / median / max, and the whole process's wall time (context tables, JSON keys, pinned buffers, HIP start-up: paid once
per process).  Last: `serverRound --rounds` with 10 and 40 rounds over the same inputs in ONE process, timed from outside
(whole process) and per round (wall).  encryptModelWeights packs the layer's mean and std_dev as two more ciphertexts: 18 per client.
usage: python tools/e2e_server_round.py [--dir /dev/shm/mkckks_e2e] [--clients 8] [--cts 16] [--keep]"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ppqsflhe_amd", "host", "build")


def run(prog, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(BIN, prog), *map(str, args)], capture_output=True, text=True, env=e)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise SystemExit(f"{prog} failed ({r.returncode})")
    return r, dt


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/dev/shm/mkckks_e2e")
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--cts", type=int, default=16)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, nargs="*", default=[10, 40], help="rounds per process for the --rounds legs")
    ap.add_argument("--arms", default="", help="instead of the default arms: 'K=V,K=V;K=V;...' (environment of serverRound per arm)")
    args = ap.parse_args()
    d = args.dir
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    p = lambda name: os.path.join(d, name)  # noqa: E731
    try:
        with open(p("config_cc.json"), "w") as f:
            json.dump({"MultiplicativeDepth": 10, "ScalingModSize": 50, "FirstModSize": 60, "NumLargeDigits": 3,
                       "RingDim": 65536, "BatchSize": 32768, "PREMode": "INDCPA"}, f)
        run("genCC", p("config_cc.json"), p("CC.json"))
        C, B = args.clients, args.cts
        rng = np.random.default_rng(5)
        t_prep = time.perf_counter()
        run("keyGen", p("CC.json"), p("pkT"), p("skT"))
        pairs = []
        mean = np.zeros(B * 32768)
        for c in range(C):
            run("keyGen", p("CC.json"), p(f"pk{c}"), p(f"sk{c}"))
            vals = rng.uniform(-0.5, 0.5, B * 32768)
            mean += vals / C
            with open(p(f"w{c}.json"), "w") as f:
                json.dump({"weights_summary": [{"layer": "dense", "shape": [len(vals)], "mean": float(vals.mean()),
                                                "std_dev": float(vals.std()), "values": vals.tolist()}]}, f)
            run("encryptModelWeights", p("CC.json"), p(f"pk{c}"), p(f"w{c}.json"), p(f"enc{c}.mkws"))
            os.remove(p(f"w{c}.json"))
            run("REkeyGen", p("CC.json"), p(f"sk{c}"), p("pkT"), p(f"rk{c}"))
            pairs += [p(f"rk{c}"), p(f"enc{c}.mkws")]
        t_prep = time.perf_counter() - t_prep
        enc_mb = os.path.getsize(p("enc0.mkws")) / 1048576.0
        rk_mb = os.path.getsize(p("rk0")) / 1048576.0
        print(f"# serverRound at the C3 shape: {C} clients x {B} ciphertexts, N=2^16, L=12, dnum=3; files in {d}")
        print(f"# input: {enc_mb:.1f} MiB of ciphertexts + {rk_mb:.1f} MiB re-encryption key per client; preparing them "
              f"(keyGen/encrypt/REkeyGen, {C} clients) took {t_prep:.1f} s")
        arms = [(f"pipelined, {t:2d} I/O threads", {"MKCKKS_IO_THREADS": str(t)}) for t in (1, 2, 4, 8, 12)]
        arms += [(f"pipelined, 8 threads, chunk {c:2d}", {"MKCKKS_IO_THREADS": "8", "MKCKKS_ROUND_CHUNK": str(c)}) for c in (1, 3, 6, 18)]
        arms.append(("synchronous (MKCKKS_SYNC_IO=1)", {"MKCKKS_SYNC_IO": "1"}))
        if args.arms:
            arms = [(a, dict(kv.split("=", 1) for kv in a.split(",") if kv)) for a in args.arms.split(";")]
        digest = None
        n_total = None
        print(f"{'arm':34s} {'round ms: min / median / max':>30s} {'ct/s (median)':>14s} {'process wall s':>15s}   last line of the median run")
        for name, env in arms:
            runs = []
            for rep in range(args.reps):
                out = p("agg.mkws")
                if os.path.exists(out):
                    os.remove(out)
                r, dt = run("serverRound", p("CC.json"), out, *pairs, env=env)
                line = [ln for ln in r.stdout.splitlines() if "[round] timing:" in ln]
                ms = None
                if line:
                    m = re.search(r"(\d+) clients x (\d+) ciphertexts.*files to file ([0-9.]+) ms", line[0])
                    n_total, ms = int(m.group(1)) * int(m.group(2)), float(m.group(3))
                runs.append((ms if ms is not None else dt * 1e3, dt, line[0] if line else ""))
                h = sha(out)
                if digest is None:
                    digest = h
                if h != digest:
                    raise SystemExit(f"arm '{name}' wrote different bytes")
            runs.sort()
            med = runs[len(runs) // 2]
            walls = sorted(r[1] for r in runs)
            if med[2]:
                print(f"{name:34s} {runs[0][0]:9.1f} / {med[0]:6.1f} / {runs[-1][0]:6.1f} {n_total / med[0] * 1e3:14.0f} {walls[len(walls) // 2]:15.2f}")
                print("    " + med[2].strip())
            else:
                print(f"{name:34s} {'(whole process:)':>30s} {(n_total or C * B) / walls[len(walls) // 2]:14.0f} {walls[len(walls) // 2]:15.2f}")
        # one process, many rounds (serverRound --rounds): context, keys, buffers and resolved kernels stay
        for n_rounds in args.rounds:
            with open(p("rounds.txt"), "w") as f:
                for r in range(n_rounds):
                    f.write(" ".join([p(f"agg_r{r}.mkws"), *pairs]) + "\n")
            r, dt = run("serverRound", p("CC.json"), "--rounds", p("rounds.txt"), env={"MKCKKS_IO_THREADS": "8"})
            per = [float(m) for m in re.findall(r"round \d+ of \d+: ([0-9.]+) ms wall", r.stdout)]
            last = [ln for ln in r.stdout.splitlines() if " rounds, " in ln][-1]
            for k in range(n_rounds):
                if sha(p(f"agg_r{k}.mkws")) != digest:
                    raise SystemExit(f"round {k} of the {n_rounds}-round process wrote different bytes")
                os.remove(p(f"agg_r{k}.mkws"))
            print(f"one process, {n_rounds:3d} rounds (8 I/O threads): whole process {dt:6.2f} s = {n_rounds * n_total / dt:6.0f} ct/s; per round "
                  f"wall ms: first {per[0]:.1f}, then min {min(per[1:]):.1f} / median {sorted(per[1:])[len(per[1:]) // 2]:.1f} / max {max(per[1:]):.1f}")
            print("    " + last.strip())
        # the aggregate, decrypted under the target key, is the plaintext mean (the size-independent check of the whole chain
        # at this shape: encode, encrypt, 8 x 18 re-encryptions, sum, rescale * 1/8, decrypt, decode)
        run("decryptModelWeights", p("CC.json"), p("skT"), p("agg.mkws"), p("dec.json"))
        dec = np.array(json.load(open(p("dec.json")))["weights_summary"][0]["values"])
        err = float(np.abs(dec - mean).max())
        if dec.size != mean.size or not err < 2.0 ** -25:
            raise SystemExit(f"decrypted aggregate differs from the plaintext mean: {dec.size} values, max error {err:g}")
        print(f"# decrypted aggregate = plaintext mean of the {C} clients: {dec.size} values, max |error| {err:.3g}")
        print(f"# every arm wrote the same aggregate: sha256 {digest[:16]}..., {os.path.getsize(p('agg.mkws')) / 1048576.0:.1f} MiB")
    finally:
        if not args.keep:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
