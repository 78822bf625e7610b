#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path when the caller holds HOST buffers (never `value` of bench.py):
upload 8 clients x 16 ciphertexts (C3), reencrypt_sum + rescale_mult_const on the device, download the 16 results.
Two variants: the C-ABI's synchronous mkckks_upload/download from pageable numpy memory, and pinned torch buffers
copied with non_blocking=True on the same stream (what a host integration with pinned staging would do)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppqsflhe_amd import Context

C, B = 8, 16
ctx = Context(16, 10, 50, 60, dnum=3, device=0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
N, L, D, beta = ctx.N, ctx.L, ctx.D, ctx.beta
rng = np.random.default_rng(0)
cts = np.empty((C, B, 2, L, N), dtype=np.uint64)
for l in range(L):
    cts[:, :, :, l, :] = rng.integers(0, int(ctx.moduli[l]), size=(C, B, 2, N), dtype=np.uint64)
evk = np.empty((C, beta, 2, D, N), dtype=np.uint64)
for l in range(D):
    evk[:, :, :, l, :] = rng.integers(0, int(ctx.moduli[l]), size=(C, beta, 2, N), dtype=np.uint64)
d_evk = ctx.to_device(evk)          # re-encryption keys stay resident (uploaded once per round)
d_cts = ctx.empty(cts.shape)
d_agg = ctx.empty((B, 2, L, N))
d_out = ctx.empty((B, 2, L - 1, N))
for tag in ("warm", "pageable"):
    t0 = time.perf_counter()
    d_cts.upload(cts)
    ctx.reencrypt_sum(d_cts, d_evk, d_agg, C, B, L)
    ctx.rescale_mult_const(d_agg, d_out, B, L, 1.0 / C)
    out = d_out.to_host()
    dt = time.perf_counter() - t0
    if tag != "warm":
        print(f"pageable host buffers, synchronous mkckks_upload/download: {C * B / dt:8.0f} ct/s  ({dt * 1e3:.1f} ms per {C * B} ct, "
              f"{cts.nbytes / 2**30:.2f} GiB up, {out.nbytes / 2**20:.0f} MiB down)")
h_in = torch.from_numpy(cts.view(np.int64)).pin_memory()
h_out = torch.empty((B, 2, L - 1, N), dtype=torch.int64).pin_memory()
t_in = torch.empty(cts.shape, dtype=torch.int64, device="cuda")
t_agg = torch.empty((B, 2, L, N), dtype=torch.int64, device="cuda")
t_out = torch.empty((B, 2, L - 1, N), dtype=torch.int64, device="cuda")
t_evk = torch.from_numpy(evk.view(np.int64)).cuda()
for tag in ("warm", "pinned"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_in.copy_(h_in, non_blocking=True)
    ctx.reencrypt_sum(t_in, t_evk, t_agg, C, B, L)
    ctx.rescale_mult_const(t_agg, t_out, B, L, 1.0 / C)
    h_out.copy_(t_out, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if tag != "warm":
        print(f"pinned host buffers, async copies on the compute stream:  {C * B / dt:8.0f} ct/s  ({dt * 1e3:.1f} ms per {C * B} ct)")
assert np.array_equal(h_out.numpy().view(np.uint64), out)
ctx.close()
