"""Data-parallel plumbing: one process per GPU, ranks joined through RCCL inside the engine.

The host side only has to (1) hand rank 0's RCCL unique id to every rank and (2) split a global
batch / an inference set into contiguous per-rank shards.  Any object with
broadcast_object_list (torch.distributed with the gloo backend) can carry the id.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

from . import engine as E


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of n items owned by `rank`: sizes differ by at most one, earlier ranks larger."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def loss_normalisers(global_batch: int, stamp_elems: int, kl_weight: float, kl_multiplicity: int):
    """Factors a shard applies so that per-shard sums ADD UP to the single-device Keras losses:
    nll_mean = sum(nll) / (B_global * H*W*C);  kl_reg = k * w * sum_b(KL_b) / B_global^2."""
    return 1.0 / (global_batch * stamp_elems), kl_multiplicity * kl_weight / float(global_batch) ** 2


def exchange_unique_id(rank: int, dist) -> bytes:
    """Rank 0 creates the RCCL unique id; every rank returns the same 128 bytes."""
    payload = [E.Context.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(payload, src=0)
    return payload[0]


def make_context(rank: int, world: int, local_rank: Optional[int] = None, dist=None) -> E.Context:
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", rank))
    if world == 1:
        return E.Context(local_rank, 0, 1, None)
    if dist is None:
        raise ValueError("world > 1 needs a torch.distributed-like object to exchange the RCCL id")
    uid = exchange_unique_id(rank, dist)
    return E.Context(local_rank, rank, world, uid)
