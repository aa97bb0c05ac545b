"""Multi-GPU side of the decode path: independent sequences are sharded across ranks (one process
per GPU) and the only exchange per decode step is one all_gather of the sampled token ids
(the reference itself has no distributed code).

``torch.distributed`` backend "nccl" is RCCL on ROCm; the same functions run on "gloo" for the
CPU tests.  No collective touches the KV path.
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_sequences(n_seqs: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block assignment ``[begin, end)`` of sequence ids to ``rank``; the first
    ``n_seqs % world_size`` ranks take one extra.  Equal-length sequences need no balancing."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world {world_size}")
    base, extra = divmod(n_seqs, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_by_cost(costs: List[float], world_size: int) -> List[List[int]]:
    """Ragged lengths: longest-processing-time assignment of sequence ids to ranks.  ``costs[i]`` is
    the per-step byte cost of sequence i (min(pages, budget) KV pages + pages/page_size metadata)."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: loads[j])
        out[r].append(i)
        loads[r] += costs[i]
    for lst in out:
        lst.sort()
    return out


def gather_tokens(local_tokens: torch.Tensor, dist) -> torch.Tensor:
    """One fused all_gather of this rank's sampled token ids (int64 ``[seqs_per_rank]``) ->
    ``[world * seqs_per_rank]`` in rank order.  Latency-bound (64 B per rank at 8 seqs/GPU)."""
    world = dist.get_world_size()
    out = torch.empty(world * local_tokens.numel(), dtype=local_tokens.dtype, device=local_tokens.device)
    dist.all_gather_into_tensor(out, local_tokens.contiguous())
    return out
