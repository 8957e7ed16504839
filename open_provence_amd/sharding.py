"""Batch-of-pairs data parallelism: one process per GPU, RCCL gather of the per-pair outputs.

The reference has no multi-GPU path (no collective call sites).  Every (query, block) row
of ``process()`` is an independent forward (``standalone.py:2748-2756`` builds independent jobs), so the
path shards by rows with a full weight replica per GPU and needs exactly ONE exchange step: gathering
each row's ranking logits and per-token pruning logits on the rank that post-processes.  There is no
all-reduce anywhere.

* :func:`partition_rows` -- deterministic token-balanced assignment (longest-first greedy), computed
  identically on every rank from the row lengths alone (no communication).
* :class:`ShardPlan` / :func:`gather_row_outputs` -- ``torch.distributed.gather`` of fixed-size padded payloads to
  ``dst`` (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests) and a vectorised restore of the
  original row order.  The payload is KB-MB, i.e. latency-bound: one direct gather (each peer uses its own xGMI link to
  the root) rather than a ring.  ``process()`` ships keep-probabilities (4 B per token) + ranking logits.
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


class ShardPlan:
    """Everything about one sharded batch that follows from the row lengths alone -- computed identically on every
    rank, without communication: which rows each rank runs, the size of the (padded) payload each rank sends, and the
    index map that puts the gathered per-token values back into the original row order with ONE indexing operation.

    Payload of a rank = ``[max_tokens * width]`` per-token values (its rows end to end) + ``[max_rows * nl]`` ranking
    logits, fp32.  ``width`` is 1 for keep-probabilities (4 B per token: what ``process()`` ships) or 2 for the raw
    pruning logits."""

    def __init__(self, lengths: Sequence[int], world_size: int, *, width: int = 1, num_labels: int = 1,
                 shards: Sequence[Sequence[int]] | None = None) -> None:
        self.lengths = np.asarray([int(n) for n in lengths], dtype=np.int64)
        self.world_size = int(world_size)
        self.width = int(width)
        self.num_labels = int(num_labels)
        if shards is None:
            self.shards = partition_rows(self.lengths.tolist(), self.world_size)
        else:  # a given assignment (every rank must pass the same one): e.g. one half of each rank's rows, see split()
            self.shards = [sorted(int(i) for i in s) for s in shards]
            seen = sorted(i for s in self.shards for i in s)
            if len(self.shards) != self.world_size or seen != list(range(len(self.lengths))):
                raise ValueError("shards must assign every row to exactly one of world_size ranks")
        self.tokens = [int(self.lengths[s].sum()) if s else 0 for s in self.shards]
        self.rows = [len(s) for s in self.shards]
        self.max_tokens = max(self.tokens + [1])
        self.max_rows = max(self.rows + [1])
        self.payload_size = self.max_tokens * self.width + self.max_rows * self.num_labels
        n = len(self.lengths)
        self.cu = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(self.lengths, out=self.cu[1:])
        # where row i starts inside the stacked [world, payload_size] bucket (in units of tokens), and its rank slot
        start = np.zeros(n, dtype=np.int64)
        slot = np.zeros(n, dtype=np.int64)
        for r, shard in enumerate(self.shards):
            if not shard:
                continue
            idx = np.asarray(shard, dtype=np.int64)
            off = np.zeros(len(idx), dtype=np.int64)
            np.cumsum(self.lengths[idx][:-1], out=off[1:])
            start[idx] = r * self.payload_size + off * self.width
            slot[idx] = r * self.payload_size + self.max_tokens * self.width + np.arange(len(idx)) * self.num_labels
        total = int(self.cu[-1])
        token_row_start = np.repeat(start - self.cu[:-1] * self.width, self.lengths * self.width) if total else np.zeros(0, np.int64)
        self.token_index = token_row_start + np.arange(total * self.width, dtype=np.int64)
        self.rank_index = (slot[:, None] + np.arange(self.num_labels, dtype=np.int64)[None, :]).reshape(-1)

    def local_rows(self, rank: int) -> list[int]:
        return self.shards[rank]

    def split(self, parts: int = 2) -> list[tuple["ShardPlan", list[int]]]:
        """This plan cut into ``parts`` plans over disjoint subsets of the rows: part ``j`` takes the ``j``-th contiguous
        slice of EVERY rank's rows (equal row counts up to one).  Returns ``(plan_j, rows_j)`` pairs -- ``rows_j`` = the
        part's rows as indices into this plan's rows, ascending; ``plan_j`` indexes into ``rows_j``.  A rank that runs its
        rows as ``parts`` independent launch sequences gathers each part with its own plan, so no sequence ever waits
        for another (bench.py)."""

        out = []
        for j in range(parts):
            per_rank = []
            for shard in self.shards:
                n = len(shard)
                per_rank.append(shard[n * j // parts : n * (j + 1) // parts])
            rows_j = sorted(i for s in per_rank for i in s)
            pos = {row: k for k, row in enumerate(rows_j)}
            plan_j = ShardPlan(self.lengths[rows_j].tolist() if rows_j else [], self.world_size, width=self.width,
                               num_labels=self.num_labels, shards=[[pos[i] for i in s] for s in per_rank])
            out.append((plan_j, rows_j))
        return out

    def pack(self, rank: int, values: torch.Tensor, rank_logits: torch.Tensor, *, out: torch.Tensor | None = None) -> torch.Tensor:
        """This rank's payload (on the tensors' device): a NEW tensor on every call (``out`` given: written there --
        :meth:`gather` passes its private, zero-filled-once buffer)."""

        if values.numel() != self.tokens[rank] * self.width or rank_logits.numel() != self.rows[rank] * self.num_labels:
            raise ValueError("local outputs do not match this rank's shard")
        payload = out if out is not None else torch.zeros(self.payload_size, dtype=torch.float32, device=values.device)
        payload[: values.numel()].copy_(values.reshape(-1))
        base = self.max_tokens * self.width
        payload[base : base + rank_logits.numel()].copy_(rank_logits.reshape(-1))
        return payload

    def _send_buffer(self, device: torch.device, rank: int) -> torch.Tensor:
        """:meth:`gather`'s own send buffer, one per (device, rank), zero-filled ONCE: the padding behind the values /
        logits is never written again, so a step of a long run only copies its two tensors in (no allocation, no fill).
        Private to ``gather``: nothing a caller holds aliases it."""

        cache = self.__dict__.setdefault("_payload_cache", {})
        key = (str(device), int(rank))
        if key not in cache:
            cache[key] = torch.zeros(self.payload_size, dtype=torch.float32, device=device)
        return cache[key]

    def unpack(self, bucket: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """``bucket[world * payload_size]`` (the gathered payloads, rank-major) -> per-token values ``[T, width]`` and
        ranking logits ``[B, nl]`` in the ORIGINAL row order: two index_select operations, no per-row loop."""

        flat = bucket.reshape(-1)
        tok_idx, rk_idx = self._device_indices(flat.device)
        tok = flat[tok_idx].reshape(-1, self.width)
        rk = flat[rk_idx].reshape(len(self.lengths), self.num_labels)
        return tok, rk

    def _device_indices(self, device: torch.device) -> tuple[torch.Tensor, torch.Tensor]:
        """The two index maps on ``device``, uploaded once per plan: at 8 ranks x 256 pairs x 512 tokens the token map
        is 8 MB, and a pageable host-to-device copy of it inside every step would stall the root rank -- the rank whose
        time the max-over-ranks measurement takes."""

        cache = self.__dict__.setdefault("_index_cache", {})
        key = str(device)
        if key not in cache:
            cache[key] = (torch.from_numpy(self.token_index).to(device), torch.from_numpy(self.rank_index).to(device))
        return cache[key]

    def gather(
        self,
        values: torch.Tensor,
        rank_logits: torch.Tensor,
        *,
        dst: int = 0,
        group: dist.ProcessGroup | None = None,
    ) -> tuple[torch.Tensor, torch.Tensor] | None:
        """ONE ``torch.distributed.gather`` of the fixed-size payloads to ``dst`` (no size exchange: every rank derives
        the sizes from the plan).  Returns ``unpack`` of the bucket on ``dst``, ``None`` elsewhere."""

        me = dist.get_rank(group)  # rank INSIDE the group: what shards / tokens are indexed by, and what `dst` names
        payload = self.pack(me, values, rank_logits, out=self._send_buffer(values.device, me))
        # torch.distributed.gather takes the GLOBAL rank of the destination: translate for a real sub-group
        dst_global = dist.get_global_rank(group, dst) if group is not None and group is not dist.group.WORLD else dst
        if me == dst:
            cache = self.__dict__.setdefault("_bucket_cache", {})
            key = str(payload.device)
            if key not in cache:  # the root's receive bucket and its per-rank views, built once per plan
                bucket = torch.empty(self.world_size * self.payload_size, dtype=torch.float32, device=payload.device)
                cache[key] = (bucket, list(bucket.split(self.payload_size)))
            bucket, views = cache[key]
            dist.gather(payload, gather_list=views, dst=dst_global, group=group)
            return self.unpack(bucket)
        dist.gather(payload, gather_list=None, dst=dst_global, group=group)
        return None


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
    """Gather every rank's packed ``prune[T_r, 2]`` and ``rank_logits[B_r, nl]`` on ``dst``; returns, on ``dst``, the
    per-row pruning logits (views of one tensor, original row order) and ``rank_logits[B, nl]``; ``None`` elsewhere."""

    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    nl = rank_logits.shape[1] if rank_logits.ndim == 2 else 1
    plan = ShardPlan(lengths, world, width=2, num_labels=nl, shards=[list(s) for s in shards])  # any assignment of rows to ranks
    if list(local_rows) != plan.shards[me]:
        raise ValueError("local outputs do not match this rank's shard")
    out = plan.gather(prune, rank_logits, dst=dst, group=group)
    if out is None:
        return None
    tok, all_rank = out
    return list(tok.split(plan.lengths.tolist())), all_rank


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


# ------------------------------------------------------------------------------------------------------------------
# One arithmetic per job (round 6).  Every rank chooses its kernel set at load time from the same weights on the same
# synthetic batch with deterministic kernels -- the same choice -- but the AUDIT of that choice looks at real rows, and a
# rank's shard is not its neighbour's: left to themselves, a rank that failed alone went back to the default selection
# alone (outputs depending on how the batch was cut, and one rank 1.8 x slower inside a synchronous gather).  Under a
# process group the ranks therefore (a) compare their load-time choices once, (b) audit the SAME rows and combine the
# verdicts with MIN: all keep the calibrated set or all leave it.  What is gathered afterwards: standalone.py:3075-3092.
# ``encoder`` = HipEncoder, or anything with effective_policy() / audit_pending / audit_rows(rows) / revert_to_default(reason).
# ------------------------------------------------------------------------------------------------------------------
def _flag_device(group) -> "torch.device":
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def agree_on_kernel_set(encoder, group=None) -> str:
    """Collective, once per attach: the ranks' kernel sets are compared; any difference sends EVERY rank to the default
    selection.  Returns the set all ranks run afterwards."""

    import torch.distributed as dist

    policy = encoder.effective_policy()
    mine = policy["kernel_set"]
    world = dist.get_world_size(group)
    if world <= 1:
        return mine
    # (sets 8 / 9 run layer by layer: the layers that keep the corrected MLP are part of the arithmetic)
    layers = policy.get("mlp_correction_layers")
    key = mine if layers is None else f"{mine}@{','.join(map(str, layers))}"
    names: list = [None] * world
    dist.all_gather_object(names, key, group=group)
    if any(n != names[0] for n in names):
        return encoder.revert_to_default(f"the ranks of the process group chose different kernel sets at load time: {names}")
    return mine


def collective_audit(encoder, rows, group=None) -> "bool | None":
    """Collective: every rank audits the SAME ``rows`` (a sample every rank has: the head of the request) and the verdicts are
    combined with MIN -- one failing rank sends every rank to the default selection.  No-op (and no collective) when no
    calibrated set is pending: that state is the same on every rank.  Returns the common verdict (None: nothing audited)."""

    import torch.distributed as dist

    if not encoder.audit_pending:
        return None
    verdict = encoder.audit_rows(rows)
    flag = torch.tensor([0 if verdict is False else 1], dtype=torch.int32, device=_flag_device(group))
    if dist.get_world_size(group) > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        encoder.revert_to_default("the first-batch audit failed on at least one rank of the process group")
        return False
    return None if verdict is None else True
