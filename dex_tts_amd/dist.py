"""Utterance sharding across the GPUs of one node: one process per GPU, every rank holds a
full weight replica, independent utterances are dealt round-robin by length, and the finished mels are
exchanged with ONE all-gather (RCCL over xGMI on GPUs; gloo in the CPU tests).  Nothing inside a sampler
call is collective — global reductions in GroupNorm / InstanceNorm / softmax make sequence sharding
unnatural, so anything finer than utterance granularity is "replicas only".

Padding is not neutral in the reference (norm/softmax statistics include padded columns), so the padded
length of every utterance is fixed BEFORE sharding: all ranks pad to the global batch maximum, which is
exactly what a single-GPU batched run would do.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist

from .config import fix_len_compatibility


def partition(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Deal utterance indices to ranks: sort by length (descending), round-robin (balances sum T and sum N^2)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world)]
    for pos, idx in enumerate(order):
        shards[pos % world].append(idx)
    return shards


def padded_length(lengths: Sequence[int], n_stages: int = 2) -> int:
    return fix_len_compatibility(max(int(l) for l in lengths), n_stages)


def sample_sharded(sample_fn: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                   mu: torch.Tensor, mask: torch.Tensor, z: torch.Tensor, lengths: Sequence[int],
                   group=None) -> torch.Tensor:
    """Every rank calls this with the SAME full batch (mu, mask, z: [B,80,T] / [B,1,T], T already the
    global padded length).  Each rank samples only its shard via ``sample_fn(z, mask, mu) -> [b,80,T]``
    and the results are all-gathered; returns the full [B,80,T] on every rank, in input order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = mu.shape[0]
    shards = partition(lengths, world)
    per = max(len(s) for s in shards)
    mine = shards[rank]
    out_local = torch.zeros(per, mu.shape[1], mu.shape[2], dtype=torch.float32, device=mu.device)
    if mine:
        idx = torch.as_tensor(mine, device=mu.device)
        out_local[: len(mine)] = sample_fn(z.index_select(0, idx), mask.index_select(0, idx), mu.index_select(0, idx))
    if world == 1:
        gathered = out_local[None]
    else:
        gathered = torch.empty(world, *out_local.shape, dtype=out_local.dtype, device=out_local.device)
        dist.all_gather_into_tensor(gathered.view(-1, *out_local.shape[1:]), out_local, group=group)
    full = torch.empty(B, mu.shape[1], mu.shape[2], dtype=torch.float32, device=mu.device)
    for r, s in enumerate(shards):
        for slot, i in enumerate(s):
            full[i] = gathered[r, slot]
    return full
