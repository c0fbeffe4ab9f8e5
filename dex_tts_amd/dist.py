"""Utterance sharding across the GPUs of one node: one process per GPU, every rank holds a
full weight replica, independent utterances are dealt round-robin by length, and the finished mels are
exchanged with ONE all-gather (RCCL over xGMI on GPUs; gloo in the CPU tests).  Nothing inside a sampler
call is collective — global reductions in GroupNorm / InstanceNorm / softmax make sequence sharding
unnatural, so anything finer than utterance granularity is "replicas only".

Padding is not neutral in the reference (norm/softmax statistics include padded columns), so the padded
length of every utterance is fixed BEFORE sharding: all ranks pad to the global batch maximum, which is
exactly what a single-GPU batched run would do.

Opt-in LENGTH BUCKETING (``sample_bucketed``): padding to the global maximum makes every kernel process the padding too (with
lengths spread over 0.6 T .. T about a fifth of the frames of a batch are padding, and the metric counts valid frames).  A bucket
is the set of utterances whose length rounds up to the same multiple of ``bucket_width`` frames; each bucket is padded to ITS OWN
maximum and sampled as a batch of its own, i.e. its result is exactly what the reference computes when it is handed that bucket
as its batch ("pad to the global batch max, or to per-bucket max").  It is NOT the result of the globally padded
batch - padded columns enter the norm / softmax statistics - which is why bucketing is never the default.

A rank needs only ITS utterances plus the lengths of all of them (``take_shard`` / ``local=True``): the
partition is a pure function of the lengths, so every rank derives the same deal without communication.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

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


def _world_rank(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def take_shard(t, lengths: Sequence[int], group=None):
    """Rows of a full-batch tensor (or of every tensor of a list) that belong to this rank, in shard order (what
    ``local=True`` expects)."""
    if isinstance(t, (list, tuple)):
        return [take_shard(u, lengths, group) for u in t]
    world, rank = _world_rank(group)
    idx = torch.as_tensor(partition(lengths, world)[rank], dtype=torch.long, device=t.device)
    return t.index_select(0, idx)


def _all_gather(out_local: torch.Tensor, world: int, group) -> torch.Tensor:
    """[per, ...] on every rank -> [world * per, ...].  RCCL gathers device tensors directly; the gloo backend
    (CPU tests, single-GPU smoke runs) is staged through host memory."""
    if world == 1:
        return out_local
    backend = dist.get_backend(group)
    if backend == "gloo" and out_local.is_cuda:
        host = out_local.cpu()
        g = torch.empty(world * host.shape[0], *host.shape[1:], dtype=host.dtype)
        dist.all_gather_into_tensor(g, host, group=group)
        return g.to(out_local.device)
    g = torch.empty(world * out_local.shape[0], *out_local.shape[1:], dtype=out_local.dtype, device=out_local.device)
    dist.all_gather_into_tensor(g, out_local.contiguous(), group=group)
    return g


def sample_sharded(sample_fn: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                   mu: torch.Tensor, mask: torch.Tensor, z: torch.Tensor, lengths: Sequence[int],
                   group=None, local: bool = False, extras: Optional[dict] = None) -> torch.Tensor:
    """Sample a batch of ``len(lengths)`` utterances over the ranks of ``group``.

    ``lengths`` lists ALL utterances (identical on every rank); T of the tensors is already the global padded
    length.  With ``local=False`` every rank passes the full batch (mu, z: [B,80,T], mask: [B,1,T]) and its rows are
    selected here; with ``local=True`` a rank passes only its own utterances in shard order (``take_shard``).
    Each rank runs ``sample_fn(z, mask, mu) -> [b,80,T]`` on its shard; ONE all-gather exchanges the finished mels
    and one ``index_copy_`` puts them back in input order.  Returns the full [B,80,T] on every rank.

    ``extras``: further per-utterance inputs of the sampler, name -> tensor or list of tensors with the utterance on dim 0
    (DEX: ``ref`` = the six [B,mid,Tr] TIV skips, ``sty`` [B,mid,Ts], ``sty_lengths`` [B]; GeDEX-VCTK: ``spk`` [B,64]).  They
    are sharded exactly like mu / mask / z and handed to ``sample_fn`` as keyword arguments."""
    world, rank = _world_rank(group)
    B = len(lengths)
    shards = partition(lengths, world)
    per = max(len(s) for s in shards)
    mine = shards[rank]
    extras = dict(extras or {})
    if not local and world > 1:
        idx = torch.as_tensor(mine, dtype=torch.long, device=mu.device)
        mu, mask, z = mu.index_select(0, idx), mask.index_select(0, idx), z.index_select(0, idx)
        sel = lambda t: t.index_select(0, idx.to(t.device))
        extras = {k: ([sel(t) for t in v] if isinstance(v, (list, tuple)) else sel(v)) for k, v in extras.items()}
    for k, v in extras.items():
        for t in (v if isinstance(v, (list, tuple)) else [v]):
            if t.shape[0] != len(mine):
                raise ValueError(f"rank {rank}: extras[{k!r}] holds {t.shape[0]} utterances, its shard has {len(mine)}")
    if mu.shape[0] != len(mine):
        raise ValueError(f"rank {rank} holds {mu.shape[0]} utterances, its shard has {len(mine)}")
    F, T = mu.shape[1], mu.shape[2]
    if len(mine) == per:
        out_local = sample_fn(z, mask, mu, **extras)
    else:                                   # uneven deal: pad the gather slot, not the sampler batch
        out_local = torch.zeros(per, F, T, dtype=torch.float32, device=mu.device)
        if mine:
            out_local[: len(mine)] = sample_fn(z, mask, mu, **extras)
    if world == 1 and not local:
        return out_local                    # full batch in, sampled in input order: nothing to exchange or permute
    # (world == 1 with local=True: the caller handed the rows in SHARD order — length-sorted — so the un-permute below still runs)
    gathered = _all_gather(out_local, world, group)                     # [world * per, F, T], slot r*per + s
    src = [r * per + s for r, sh in enumerate(shards) for s in range(len(sh))]
    dst = [i for sh in shards for i in sh]
    full = torch.empty(B, F, T, dtype=torch.float32, device=mu.device)
    dst_t = torch.as_tensor(dst, dtype=torch.long, device=mu.device)
    if len(src) == world * per:
        full.index_copy_(0, dst_t, gathered)
    else:
        full.index_copy_(0, dst_t, gathered.index_select(0, torch.as_tensor(src, dtype=torch.long, device=mu.device)))
    return full


def buckets_of(lengths: Sequence[int], bucket_width: int, n_stages: int = 2) -> List[tuple]:
    """[(padded_T, [utterance indices])] - utterances whose length rounds up to the same multiple of ``bucket_width`` share a
    bucket, padded to the bucket's own maximum (``fix_len_compatibility``); ascending padded length, input order inside a bucket."""
    if bucket_width < 1:
        raise ValueError("bucket_width must be >= 1")
    groups = {}
    for i, l in enumerate(lengths):
        groups.setdefault(-(-int(l) // bucket_width), []).append(i)
    return [(padded_length([lengths[i] for i in idx], n_stages), idx) for _, idx in sorted(groups.items())]


def sample_bucketed(sample_fn: Callable[..., torch.Tensor], mu: torch.Tensor, mask: torch.Tensor, z: torch.Tensor, lengths: Sequence[int],
                    bucket_width: int, group=None, extras: Optional[dict] = None, n_stages: int = 2) -> torch.Tensor:
    """``sample_sharded`` bucket by bucket (see the module docstring): every bucket is cropped to its own padded length, dealt over the
    ranks of ``group`` and sampled as one batch; returns [B, 80, T] in input order, zero beyond a bucket's padded length.  Full-batch
    inputs on every rank (``local=False`` semantics).  Per bucket the result equals ``sample_fn`` on that bucket alone."""
    B, F, T = mu.shape
    out = torch.zeros(B, F, T, dtype=torch.float32, device=mu.device)
    for Tb, idx in buckets_of(lengths, bucket_width, n_stages):
        Tb = min(Tb, T)
        ix = torch.as_tensor(idx, dtype=torch.long, device=mu.device)
        sel = lambda t: t.index_select(0, ix.to(t.device))
        ex = {k: ([sel(t) for t in v] if isinstance(v, (list, tuple)) else sel(v)) for k, v in (extras or {}).items()}
        y = sample_sharded(sample_fn, sel(mu)[:, :, :Tb].contiguous(), sel(mask)[:, :, :Tb].contiguous(), sel(z)[:, :, :Tb].contiguous(),
                           [lengths[i] for i in idx], group=group, local=False, extras=ex)
        out[ix, :, :Tb] = y
    return out
