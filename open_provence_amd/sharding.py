"""Batch-of-pairs data parallelism: one process per GPU, RCCL gather of the per-pair outputs.

The reference has no multi-GPU path (no collective call sites).  Every (query, block) row
of ``process()`` is an independent forward (``standalone.py:2748-2756`` builds independent jobs), so the
path shards by rows with a full weight replica per GPU and needs exactly ONE exchange step: gathering
each row's ranking logits and per-token pruning logits on the rank that post-processes.  There is no
all-reduce anywhere.

* :func:`partition_rows` -- deterministic token-balanced assignment (longest-first greedy), computed
  identically on every rank from the row lengths alone (no communication).
* :func:`gather_row_outputs` -- ``torch.distributed.gather`` of fixed-size padded payloads to ``dst``
  (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).  The payload is KB-MB, i.e.
  latency-bound: one direct gather (each peer uses its own xGMI link to the root) rather than a ring.
* :func:`sharded_forward` -- partition, run the local shard through a caller-supplied forward, gather,
  and restore the original row order on ``dst``.
"""

from __future__ import annotations

from typing import Callable, Sequence

import numpy as np
import torch
import torch.distributed as dist

RowForward = Callable[[list[list[int]]], tuple[torch.Tensor, torch.Tensor, np.ndarray]]


def partition_rows(lengths: Sequence[int], world_size: int) -> list[list[int]]:
    """Row indices per rank.  Longest-processing-time greedy on token counts; ties broken by row index and
    rank index so that every rank computes the same answer.  Each rank's list is sorted ascending."""

    if world_size <= 0:
        raise ValueError("world_size must be positive")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    counts = [0] * world_size
    shards: list[list[int]] = [[] for _ in range(world_size)]
    for idx in order:
        rank = min(range(world_size), key=lambda r: (loads[r], counts[r], r))
        shards[rank].append(idx)
        loads[rank] += int(lengths[idx])
        counts[rank] += 1
    return [sorted(s) for s in shards]


def gather_row_outputs(
    prune: torch.Tensor,
    rank_logits: torch.Tensor,
    local_rows: Sequence[int],
    lengths: Sequence[int],
    shards: Sequence[Sequence[int]],
    *,
    dst: int = 0,
    group: dist.ProcessGroup | None = None,
) -> tuple[list[torch.Tensor], torch.Tensor] | None:
    """Gather every rank's packed ``prune[T_r, 2]`` and ``rank_logits[B_r, nl]`` on ``dst``.

    Buffer sizes follow from ``shards``/``lengths`` (known on every rank), so no size exchange is needed:
    each rank pads its payload to the largest shard and one ``gather`` moves it.  Returns, on ``dst``, the
    per-row pruning logits (original row order) and ``rank_logits[B, nl]``; ``None`` elsewhere."""

    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    if len(shards) != world:
        raise ValueError("shards must have one entry per rank")
    tokens = [sum(int(lengths[i]) for i in shard) for shard in shards]
    rows = [len(shard) for shard in shards]
    max_tokens, max_rows = max(tokens + [1]), max(rows + [1])
    nl = rank_logits.shape[1] if rank_logits.ndim == 2 else 1
    if prune.shape[0] != tokens[me] or rank_logits.shape[0] != rows[me] or list(local_rows) != list(shards[me]):
        raise ValueError("local outputs do not match this rank's shard")

    payload = torch.zeros(max_tokens * 2 + max_rows * nl, dtype=torch.float32, device=prune.device)
    payload[: tokens[me] * 2] = prune.reshape(-1)
    payload[max_tokens * 2 : max_tokens * 2 + rows[me] * nl] = rank_logits.reshape(-1)
    if me == dst:
        bucket = [torch.empty_like(payload) for _ in range(world)]
        dist.gather(payload, gather_list=bucket, dst=dst, group=group)
    else:
        dist.gather(payload, gather_list=None, dst=dst, group=group)
        return None

    n_rows = len(lengths)
    per_row: list[torch.Tensor | None] = [None] * n_rows
    all_rank = torch.zeros((n_rows, nl), dtype=torch.float32, device=prune.device)
    for r, shard in enumerate(shards):
        buf = bucket[r]
        p = buf[: tokens[r] * 2].reshape(tokens[r], 2)
        rk = buf[max_tokens * 2 : max_tokens * 2 + rows[r] * nl].reshape(rows[r], nl)
        cursor = 0
        for j, idx in enumerate(shard):
            n = int(lengths[idx])
            per_row[idx] = p[cursor : cursor + n]
            cursor += n
            all_rank[idx] = rk[j]
    return [t if t is not None else prune.new_zeros((0, 2)) for t in per_row], all_rank


def sharded_forward(
    rows: Sequence[Sequence[int]],
    forward_rows: RowForward,
    *,
    dst: int = 0,
    group: dist.ProcessGroup | None = None,
) -> tuple[list[torch.Tensor], torch.Tensor] | None:
    """Run ``rows`` across all ranks of ``group``.  ``forward_rows(local_rows) -> (prune[T,2], rank[B,nl], cu)``
    is normally ``HipEncoder.forward_rows``.  Every rank must pass the same ``rows``."""

    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    lengths = [len(r) for r in rows]
    shards = partition_rows(lengths, world)
    mine = shards[me]
    prune, rank_logits, _cu = forward_rows([list(rows[i]) for i in mine])
    return gather_row_outputs(prune, rank_logits, mine, lengths, shards, dst=dst, group=group)
