"""Host-side packing between the reference's padded ``[B, L]`` batches and the unpadded layout.

The reference right-pads every batch to its own maximum length (``standalone.py:2832-2880``) and HF
ModernBERT uses ``position_ids = arange(L)`` for every row (``modeling_modernbert.py:448-449``), so a
row's real tokens sit at positions ``0..len-1``.  The HIP path therefore takes only the
``attention_mask == 1`` tokens, laid end to end, plus ``cu_seqlens`` (prefix offsets); values at
padding positions are never read by ``process()`` and are returned as zeros.
"""

from __future__ import annotations

import itertools
from typing import Sequence

import numpy as np
import torch


def pack_rows(rows: Sequence[Sequence[int]]) -> tuple[np.ndarray, np.ndarray, int]:
    """Token-id rows -> (ids[int32, T], cu_seqlens[int32, B+1], max_len)."""

    lengths = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    cu = np.zeros(len(rows) + 1, dtype=np.int32)
    if len(rows):
        total = int(lengths.sum())
        if total >= 2**31:
            raise ValueError("batch has more than 2^31 tokens")
        np.cumsum(lengths, out=cu[1:])
    # one pass over all tokens in C (a per-row np.asarray + slice assignment costs 1.7x as much on ~100-token rows)
    ids = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int32, count=int(cu[-1]))
    max_len = int(lengths.max()) if len(rows) else 0
    return ids, cu, max_len


def lengths_from_mask(attention_mask: torch.Tensor) -> np.ndarray:
    """Row lengths of a right-padded mask; raises if a row is not ``1...10...0``."""

    mask = attention_mask.detach().to("cpu").to(torch.int64)
    if mask.ndim != 2:
        raise ValueError("attention_mask must be [B, L]")
    lengths = mask.sum(dim=1)
    width = mask.shape[1]
    expected = (torch.arange(width)[None, :] < lengths[:, None]).to(torch.int64)
    if not torch.equal(mask.ne(0).to(torch.int64), expected):
        raise NotImplementedError(
            "attention_mask must be right-padded (ones then zeros): the packed HIP path derives positions "
            "from token order, exactly what the reference's process() produces (standalone.py:2832-2880)."
        )
    return lengths.numpy()


def pack_padded(input_ids: torch.Tensor, attention_mask: torch.Tensor | None) -> tuple[np.ndarray, np.ndarray, int]:
    """Padded ``[B, L]`` ids + right-padded mask -> (ids[int32, T], cu_seqlens[int32, B+1], max_len)."""

    ids_cpu = input_ids.detach().to("cpu")
    if ids_cpu.ndim != 2:
        raise ValueError("input_ids must be [B, L]")
    batch, width = ids_cpu.shape
    if attention_mask is None:
        lengths = np.full(batch, width, dtype=np.int64)
    else:
        if tuple(attention_mask.shape) != (batch, width):
            raise ValueError("attention_mask shape must match input_ids")
        lengths = lengths_from_mask(attention_mask)
    cu = np.zeros(batch + 1, dtype=np.int32)
    np.cumsum(lengths, out=cu[1:])
    ids_np = ids_cpu.to(torch.int32).numpy()
    keep = np.arange(width)[None, :] < lengths[:, None]
    packed = np.ascontiguousarray(ids_np[keep], dtype=np.int32)
    return packed, cu, int(lengths.max()) if batch else 0


def unpack_to_padded(packed: torch.Tensor, cu: np.ndarray, width: int) -> torch.Tensor:
    """``[T, C]`` packed values -> ``[B, width, C]`` with zeros at padding positions (same device)."""

    batch = len(cu) - 1
    out = packed.new_zeros((batch, width) + tuple(packed.shape[1:]))
    if batch == 0 or packed.shape[0] == 0:
        return out
    lengths = np.diff(cu)
    row_index = np.repeat(np.arange(batch), lengths)
    col_index = np.arange(int(cu[-1])) - np.repeat(cu[:-1], lengths)
    ri = torch.from_numpy(row_index).to(packed.device)
    ci = torch.from_numpy(col_index).to(packed.device)
    out[ri, ci] = packed
    return out
