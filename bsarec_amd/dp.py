"""Data-parallel pieces of the training path (new functionality: the reference is single-device,
src/main.py:19).  One process per GPU; user-sequence minibatches shard across ranks; the only
exchange is ONE summing all-reduce of the flat fp32 gradient arena per step (RCCL over xGMI when the
backend is "nccl"), after which every rank runs the same fused Adam on grad_sum / world.

Loss semantics: the reference's loss is the mean CE over the batch.  With equal shards of size B the
mean over the global batch of W*B samples equals the mean of the W local means, so each rank
back-propagates its local mean and the all-reduced sum is scaled by 1/W.  A short last global batch
would break the equal-shard premise and is dropped on every rank (DeviceBatches, world > 1).
"""
from __future__ import annotations

import torch


def shard_of_global_batch(index: torch.Tensor, batch_size: int, rank: int, world: int) -> torch.Tensor:
    """Rank's slice of one global batch of world*batch_size sample indices (same permutation on every rank)."""
    return index[rank * batch_size:(rank + 1) * batch_size]


def allreduce_sum_(flat_grads: torch.Tensor, group=None, force: bool = False) -> float:
    """In-place summing all-reduce of the flat gradient arena; returns the scale (1/world) that the
    fused Adam applies to the sum.  ``force`` issues the collective even in a 1-rank group (tests, bench --dp:
    exercises the RCCL launch and its graph capture on a single GPU)."""
    world = torch.distributed.get_world_size(group)
    if world > 1 or force:
        torch.distributed.all_reduce(flat_grads, op=torch.distributed.ReduceOp.SUM, group=group)
    return 1.0 / world
