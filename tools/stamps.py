#!/usr/bin/env python3
"""In-kernel phase timing of the hot kernels (diagnostic build).  Build: make -C ppqsflhe_amd/csrc OUT=../libmkckks_stamp.so
CXXFLAGS="... -DMK_STAMP=1"; run on the GPU box:  MKCKKS_LIB=$PWD/ppqsflhe_amd/libmkckks_stamp.so MKCKKS_STAMPS=1 python tools/stamps.py
Every wave of k_conv_col (fp64-class / integer-class targets) and k_qsum3_fp stamps the shader clock at its phase
boundaries; this prints the median, 10th and 90th percentile of every phase in shader cycles, and the wave lifetime.
The stamps serialise the schedule at each mark (sched_barrier + s_waitcnt lgkmcnt(0)), so the instrumented kernels run
a few percent longer than the product ones."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ppqsflhe_amd import Context, binding
    dev = torch.device("cuda", 0)
    g = Context(16, 10, 50, 60, dnum=3, device=0)
    lib = binding.load_library()
    lib.mkckks_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t)]
    C, B, L, N, D = 8, 16, g.L, g.N, g.D
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)

    def uniform(lead, ids):
        t = torch.empty(*lead, len(ids), N, dtype=torch.int64, device=dev)
        for j, l in enumerate(ids):
            t[..., j, :] = torch.randint(0, int(g.moduli[l]), (*lead, N), generator=gen, device=dev, dtype=torch.int64)
        return t

    cts = uniform((C, B), list(range(L)) * 2).view(C, B, 2, L, N)
    evks = uniform((C,), list(range(D)) * (2 * g.beta)).view(C, g.beta, 2, D, N)
    out = torch.empty(B, 2, L, N, dtype=torch.int64, device=dev)
    g.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        g.reencrypt_sum(cts, evks, out, C, B, L)
    torch.cuda.synchronize()
    buf = np.zeros(1 << 20, dtype=np.uint64)
    names = {0: ("k_conv_col, fp64-class targets (last launch of the step)", ["conversion (loads + MACs)", "round A", "exchange + barrier",
                                                                              "round B", "stores issued"], True),
             1: ("k_conv_col, integer-class targets (last launch of the step)", ["conversion (loads + MACs)", "round A",
                                                                                 "exchange + barrier", "round B", "stores issued"], True),
             2: ("k_qsum3_fp (sums over the client loop)", ["own-digit products", "transforms", "digit products",
                                                            "conversions + store", "whole loop"], False)}
    for region, (title, phases, diffs) in names.items():
        n = ctypes.c_size_t(0)
        rc = lib.mkckks_debug_stamps(g._h, buf.ctypes.data, region, ctypes.byref(n))
        if rc != 0 or n.value == 0:
            print(f"region {region}: no stamps (product build, or MKCKKS_STAMPS not set)")
            continue
        a = buf.reshape(-1, 8).astype(np.int64)
        a = a[(a != 0).any(axis=1)]
        print(f"== {title}: {len(a)} waves")
        for i, ph in enumerate(phases):
            col = a[:, i]
            col = col[col > 0]
            if len(col):
                print(f"   {ph:28s} median {np.median(col):9.0f}  p10 {np.percentile(col, 10):9.0f}  p90 {np.percentile(col, 90):9.0f} cycles")
        if diffs:
            life = a[:, :len(phases)].sum(axis=1)
            print(f"   {'wave lifetime (stamped part)':28s} median {np.median(life):9.0f}  p10 {np.percentile(life, 10):9.0f}  p90 {np.percentile(life, 90):9.0f} cycles")
    g.close()


if __name__ == "__main__":
    main()
