"""N > 1 path on CPU: world_size-2 gloo processes run partition -> local forward -> gather -> reorder.
The local forward is a deterministic stand-in (test infrastructure); the product passes HipEncoder.forward_rows."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from open_provence_amd.sharding import partition_rows, sharded_forward


def test_partition_is_balanced_deterministic_and_complete():
    rng = np.random.default_rng(0)
    lengths = rng.choice([128, 256, 384, 512, 768, 1024, 1536, 2048], size=97).tolist()
    for world in (1, 2, 4, 8):
        shards = partition_rows(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(97))
        loads = [sum(lengths[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= 2048
        assert shards == partition_rows(lengths, world)
    assert partition_rows([], 4) == [[], [], [], []]
    assert partition_rows([5, 5, 5], 2) == [[0, 2], [1]]
    with pytest.raises(ValueError):
        partition_rows([1], 0)


def _fake_forward(rows):
    cu = np.zeros(len(rows) + 1, dtype=np.int32)
    for i, r in enumerate(rows):
        cu[i + 1] = cu[i] + len(r)
    flat = torch.tensor([t for r in rows for t in r], dtype=torch.float32)
    prune = torch.stack([flat * 0.5, flat + 1.0], dim=-1) if len(flat) else torch.zeros((0, 2))
    rank = torch.tensor([[float(sum(r)) / 7.0] for r in rows], dtype=torch.float32).reshape(len(rows), 1)
    return prune, rank, cu


def _worker(rank, world, port, rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        result = sharded_forward(rows, _fake_forward, dst=0)
        if rank == 0:
            per_row, all_rank = result
            torch.save({"per_row": per_row, "rank": all_rank}, out_path)
        else:
            assert result is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_forward_gloo_matches_single_process(tmp_path, world):
    rng = np.random.default_rng(1)
    rows = [rng.integers(1, 100, size=int(n)).tolist() for n in rng.integers(1, 40, size=11)]
    rows[4] = []  # an empty row must survive the round trip
    out_path = str(tmp_path / "out.pt")
    mp.spawn(_worker, args=(world, _free_port(), rows, out_path), nprocs=world, join=True)
    got = torch.load(out_path)
    prune, rank, cu = _fake_forward(rows)
    assert torch.equal(got["rank"], rank)
    for i in range(len(rows)):
        assert torch.equal(got["per_row"][i], prune[cu[i] : cu[i + 1]]), i
