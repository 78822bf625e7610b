"""Multi-GPU aggregation step: clients are sharded over the ranks of one node (way 2).

Each rank PREs and sums its own clients (canonical residues, < q_i < 2^61), then the per-rank partial sums are
combined with ONE integer reduce-scatter (RCCL `ncclSum` on 64-bit words over xGMI): up to 8 addends below 2^61
cannot wrap 2^64, so the unreduced sum followed by one `mod q_i` kernel (mkckks_reduce_mod_batch) equals the
coefficient-wise modular sum.  Rank r ends up owning ciphertexts [r*B/W, (r+1)*B/W) of the aggregate and
rescales / re-encrypts only those.  torch.distributed carries the collective (backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests, where reduce_scatter_tensor is emulated with all_reduce + slice).
"""
import torch
import torch.distributed as dist

MAX_TERMS = 8  # q_i < 2^61  =>  8 * q_i < 2^64


def shard_range(n_items, rank, world):
    if n_items % world:
        raise ValueError(f"{n_items} ciphertexts do not split evenly over {world} ranks")
    per = n_items // world
    return rank * per, (rank + 1) * per


def reduce_partial_sums(partial, out_shard=None, group=None):
    """partial: int64 tensor [B, 2, L, N] of canonical residues (uint64 bit patterns) on every rank.
    Returns this rank's [B/W, 2, L, N] slice of the UNREDUCED integer sum over ranks (wraps mod 2^64 exactly like
    unsigned addition; the caller applies mkckks_reduce_mod_batch / `% q`)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world > MAX_TERMS:
        raise ValueError("integer-sum collective is exact for at most 8 ranks")
    lo, hi = shard_range(partial.shape[0], rank, world)
    if out_shard is None:
        out_shard = torch.empty((hi - lo,) + tuple(partial.shape[1:]), dtype=partial.dtype, device=partial.device)
    if dist.get_backend(group) == "gloo":
        tmp = partial.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        out_shard.copy_(tmp[lo:hi])
    else:
        dist.reduce_scatter_tensor(out_shard, partial, op=dist.ReduceOp.SUM, group=group)
    return out_shard
