"""Stored outputs of the CPU oracle for the full-size GPU comparisons (test infrastructure).

The `-m gpu` suite compares the HIP path with ``oracle/modernbert_oracle.py`` at the sizes bench.py times -- every pair of
256 x 512, 64 x 2048, spot pairs of the 19-25-layer models, 11 264 sentences through ``process()`` -- and until round 6
recomputed those oracle outputs on the box's 16 host threads in every run: ~55 % of the suite's 1000 s (VERDICT r5 weak 5:
977 of the driver's 1200 s).  They are functions of seeded weights and seeded rows only, so they are kept as fixtures:
``tests/golden/oracle_cache/<tag>.npz`` = the arrays + a fingerprint of everything they depend on (every weight tensor's
shape, head, tail and sum; the token rows; the oracle's own source).  A fixture whose fingerprint does not match is ignored
and the oracle runs (slower, never wrong); ``OPEN_PROVENCE_WRITE_ORACLE_CACHE=<dir>`` makes the suite write what it computes
(how the fixtures were made: one run of the GPU suite on a GPU box with that variable, then a copy into tests/golden/).
``tests/test_oracle_golden.py::test_oracle_cache_entry_is_what_the_oracle_computes`` recomputes rows of one entry on the CPU."""

from __future__ import annotations

import hashlib
import os
from pathlib import Path
from typing import Callable, Mapping, Sequence

import numpy as np

CACHE_DIR = Path(__file__).resolve().parent / "golden" / "oracle_cache"
_ORACLE_SOURCE = Path(__file__).resolve().parents[1] / "oracle" / "modernbert_oracle.py"


def fingerprint(state: Mapping, rows: Sequence[Sequence[int]] | None, extra: str = "") -> str:
    h = hashlib.sha1()
    h.update(hashlib.sha1(_ORACLE_SOURCE.read_bytes()).digest())
    for name in sorted(state):
        t = state[name].detach().reshape(-1)
        h.update(name.encode())
        h.update(str(tuple(state[name].shape)).encode())
        h.update(t[:64].float().numpy().tobytes())
        h.update(t[-64:].float().numpy().tobytes())
        h.update(f"{t.double().sum().item():.6e}".encode())  # (six digits: the sum's last bits depend on the host's thread count)
    if rows is not None:
        h.update(np.asarray([len(r) for r in rows], dtype=np.int64).tobytes())
        h.update(np.concatenate([np.asarray(r, dtype=np.int32) for r in rows]).tobytes() if len(rows) else b"")
    h.update(extra.encode())
    return h.hexdigest()


def cached(tag: str, fp: str, compute: Callable[[], tuple]) -> tuple:
    """The arrays stored under ``tag`` when their fingerprint is ``fp``; else ``compute()`` (a tuple of numpy arrays), written to
    ``$OPEN_PROVENCE_WRITE_ORACLE_CACHE/<tag>.npz`` when that variable names a directory."""

    path = CACHE_DIR / f"{tag}.npz"
    if path.exists() and not os.environ.get("OPEN_PROVENCE_ORACLE_CACHE_OFF"):
        with np.load(path) as data:
            if str(data["fingerprint"]) == fp:
                return tuple(data[f"a{i}"] for i in range(int(data["n"])))
    out = tuple(np.asarray(a) for a in compute())
    target = os.environ.get("OPEN_PROVENCE_WRITE_ORACLE_CACHE")
    if target:
        Path(target).mkdir(parents=True, exist_ok=True)
        np.savez_compressed(Path(target) / f"{tag}.npz", fingerprint=np.asarray(fp), n=np.asarray(len(out)), **{f"a{i}": a for i, a in enumerate(out)})
    return out


def oracle_rows(tag: str, state: Mapping, dims, rows: Sequence[Sequence[int]], batch: int = 32) -> tuple[np.ndarray, np.ndarray]:
    """(pruning_logits[n, Lmax, 2], ranking_logits[n, nl]) of the fp32 oracle on ``rows`` (padded), from the fixture or computed."""

    def compute():
        import torch

        from open_provence_amd.synthetic import pad_rows
        from oracle.modernbert_oracle import oracle_forward

        ids, mask = pad_rows(rows)
        prune, rank = [], []
        with torch.no_grad():
            for start in range(0, len(rows), batch):  # (in batches the CPU caches like)
                ref = oracle_forward(state, dims, ids[start: start + batch], mask[start: start + batch], attn="sdpa")
                prune.append(ref.pruning_logits.numpy())
                rank.append(ref.ranking_logits.numpy())
        return np.concatenate(prune), np.concatenate(rank)

    return cached(tag, fingerprint(state, rows), compute)
