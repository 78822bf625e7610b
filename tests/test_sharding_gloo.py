"""N>1 path on CPU: world_size-2 gloo run of the client-sharded aggregation collective (bench.py's multi-GPU step).
The arithmetic after the collective (mod q_i) is checked with numpy uint64 -- the device kernel
mkckks_reduce_mod_batch has its own GPU parity test (test_eval_add_sum_reduce)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ppqsflhe_amd.sharding import MAX_TERMS, reduce_partial_sums, shard_range

MODULI = [1152921504606748673, 1125899908022273, 1125899904679937, 557057]  # 60-, 51-, 51-, 20-bit limbs
N, B = 256, 4


def _partial(rank, n_ct=B, extreme=False):
    rng = np.random.default_rng(100 + rank)
    x = np.empty((n_ct, 2, len(MODULI), N), dtype=np.uint64)
    for i, q in enumerate(MODULI):
        x[:, :, i, :] = rng.integers(0, q, size=(n_ct, 2, N), dtype=np.uint64)
        x[:, :, i, 0] = q - 1  # extreme residues
        if extreme:  # every rank's partial sum at the top of its range: the 8-term integer sum is as large as it gets
            x[:, :, i, : N // 2] = q - 1
    return x


def _worker(rank, world, port, ret, n_ct=B, extreme=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    part = torch.from_numpy(_partial(rank, n_ct, extreme).view(np.int64))
    shard = reduce_partial_sums(part)
    lo, hi = shard_range(n_ct, rank, world)
    assert (lo, hi) == (rank * (n_ct // world), (rank + 1) * (n_ct // world))  # rank r owns the r-th block of the aggregate
    total = sum(_partial(r, n_ct, extreme).astype(object) for r in range(world))  # exact integers
    got = shard.numpy().view(np.uint64)
    ok = True
    for i, q in enumerate(MODULI):
        exp = np.array(total[lo:hi, :, i, :] % q, dtype=np.uint64)
        ok &= bool(np.array_equal(got[:, :, i, :] % np.uint64(q), exp))
    ok &= shard.shape == (n_ct // world, 2, len(MODULI), N)
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_integer_reduce_scatter_matches_modular_sum():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_eight_rank_extreme_partial_sums_and_shard_ownership():
    # BASELINE configs[3] (64 clients over 8 GPUs: 8 clients per rank, one partial sum per rank): the collective's
    # arithmetic at world size 8 with every rank's partial sum at q - 1 on half of the coefficients -- the largest
    # 8-term integer sum there is -- and rank r left with ciphertexts [r, r + 1) of the 8 aggregates.  (The hardware
    # leg, RCCL over xGMI on 8 MI355X, is the driver's to run.)
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, 8, True), nprocs=world, join=True)
    assert len(ret) == world and all(ret[r] for r in range(world)), dict(ret)


def test_eight_canonical_terms_never_wrap():
    # the property the collective relies on: 8 residues below 2^61 sum below 2^64, also through int64 wrap-around
    q = (1 << 61) - 1
    terms = np.full((MAX_TERMS, 16), q - 1, dtype=np.uint64)
    as_i64 = terms.view(np.int64).sum(axis=0, dtype=np.int64)  # what an int64 SUM collective computes
    exact = int(q - 1) * MAX_TERMS
    assert exact < 1 << 64
    assert np.all(as_i64.view(np.uint64) == np.uint64(exact))


def test_shard_range():
    assert shard_range(16, 3, 8) == (6, 8)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
