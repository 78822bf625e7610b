#!/usr/bin/env python3
"""r03 experiment (form b): two half-batches of the server step on two HIP streams, with and without
CU masks (hipExtStreamCreateWithCUMask), against the whole batch on one stream.

The step's kernels alternate between VALU-heavy phases (conversion + column pass, P-limb inner products) and phases that
wait on memory (inverse passes of c1, k_qsum3_fp).  If two independent half-batches ran their phases out of step on the
same CUs, one's memory waits could hide under the other's arithmetic.  Arms (8 clients x 16 ciphertexts per step each):
  one     : reencrypt_sum over all 16 indices on one stream (the product path)
  two     : indices 0-7 and 8-15 on two ordinary streams, each with its own context / workspace
  two-skew: the same, the second stream starts half a step late (phases out of step by construction)
  mask-x  : the two streams own disjoint CU sets: even / odd CUs of every XCD (half the chip each)
  mask-75 : stream A gets 3 of every 4 CUs, stream B the fourth (and vice versa per half: both run 8 indices)
Prints ct/s per arm; every arm's output is checked against arm `one` (bit-exact).
usage: python tools/exp_cumask.py [--steps 20]"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hip_lib():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    return ctypes.CDLL("libamdhip64.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    from ppqsflhe_amd import Context
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hip = hip_lib()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count

    def masked_stream(pred):
        words = (n_cu + 31) // 32
        mask = (ctypes.c_uint32 * words)()
        for cu in range(n_cu):
            if pred(cu):
                mask[cu // 32] |= 1 << (cu % 32)
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
        return st

    def plain_stream():
        st = ctypes.c_void_p()
        rc = hip.hipStreamCreateWithFlags(ctypes.byref(st), ctypes.c_uint(1))  # hipStreamNonBlocking
        if rc != 0:
            raise RuntimeError(f"hipStreamCreateWithFlags failed: {rc}")
        return st

    log_n, depth, sbits, dnum = 16, 10, 50, 3
    C, B = 8, 16
    ctx0 = Context(log_n, depth, sbits, 60, dnum=dnum, device=0)
    N, L, D, beta = ctx0.N, ctx0.L, ctx0.D, ctx0.beta
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)

    def uniform(lead, ids):
        t = torch.empty(*lead, len(ids), N, dtype=torch.int64, device=dev)
        for j, l in enumerate(ids):
            t[..., j, :] = torch.randint(0, int(ctx0.moduli[l]), (*lead, N), generator=gen, device=dev, dtype=torch.int64)
        return t

    ct = uniform((C, B), list(range(L)) * 2).view(C, B, 2, L, N)
    evk = uniform((C,), list(range(D)) * (2 * beta)).view(C, beta, 2, D, N)
    # half-batches as their own contiguous [C][B/2] arrays (the API takes one stride per client)
    halves = [ct[:, :B // 2].contiguous(), ct[:, B // 2:].contiguous()]
    agg = torch.empty(B, 2, L, N, dtype=torch.int64, device=dev)
    out = torch.empty(B, 2, L - 1, N, dtype=torch.int64, device=dev)
    agg_h = [torch.empty(B // 2, 2, L, N, dtype=torch.int64, device=dev) for _ in range(2)]
    out_h = [torch.empty(B // 2, 2, L - 1, N, dtype=torch.int64, device=dev) for _ in range(2)]
    ctx0.set_stream(torch.cuda.current_stream().cuda_stream)

    def run_one():
        ctx0.reencrypt_sum(ct, evk, agg, C, B, L)
        ctx0.rescale_mult_const(agg, out, B, L, 1.0 / C)

    def timed(fn, sync):
        for _ in range(3):
            fn()
        sync()
        best = None
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            sync()
            dt = (time.perf_counter() - t0) / args.steps
            best = dt if best is None else min(best, dt)
        return best

    results = []
    t = timed(run_one, torch.cuda.synchronize)
    ref = out.clone()
    results.append(("one", t, True))

    def two_stream_arm(name, streams, skew=False):
        cxs = []
        for st in streams:
            cx = Context(log_n, depth, sbits, 60, dnum=dnum, device=0)
            cx.set_stream(st.value)
            cxs.append(cx)

        def step():
            for h in range(2):
                cxs[h].reencrypt_sum(halves[h], evk, agg_h[h], C, B // 2, L)
                cxs[h].rescale_mult_const(agg_h[h], out_h[h], B // 2, L, 1.0 / C)

        def sync():
            for st in streams:
                hip.hipStreamSynchronize(st)

        if skew:  # second stream half a step behind: one extra half-batch on stream A before the timed loops
            cxs[0].reencrypt_sum(halves[0], evk, agg_h[0], C, B // 2, L)
        tt = timed(step, sync)
        ok = bool(torch.equal(torch.cat(out_h), ref))
        for cx in cxs:
            cx.close()
        results.append((name, tt, ok))

    two_stream_arm("two", [plain_stream(), plain_stream()])
    two_stream_arm("two-skew", [plain_stream(), plain_stream()], skew=True)
    two_stream_arm("mask-x", [masked_stream(lambda cu: cu % 2 == 0), masked_stream(lambda cu: cu % 2 == 1)])
    two_stream_arm("mask-75", [masked_stream(lambda cu: cu % 4 != 3), masked_stream(lambda cu: cu % 4 == 3)])
    print(f"# {C} clients x {B} ciphertexts per step, N=2^{log_n}, L={L}; {n_cu} CUs; best of 5 blocks of {args.steps} steps")
    print(f"{'arm':10s} {'ms/step':>9s} {'ct/s':>9s} {'vs one':>7s}  bit-exact")
    for name, tt, ok in results:
        print(f"{name:10s} {tt * 1e3:9.3f} {C * B / tt:9.0f} {results[0][1] / tt:7.3f}  {ok}")
    ctx0.close()


if __name__ == "__main__":
    main()
