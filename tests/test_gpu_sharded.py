"""The N-rank path of bench.py with the REAL encoder, on the one GPU a test box has (GPU).

Two processes share ``cuda:0`` and talk over a gloo group (RCCL refuses two ranks on one device): each builds its own
HipEncoder on the same weights, takes its rows from bench.exchange_plan (rows by token count -> plan over fragment counts ->
split(2) for the two launch sequences), runs them with ``forward_packed_on`` on its two pipelines, reduces the
keep-probabilities to per-fragment means on the device (``op_segment_means``) and gathers them + the ranking logits to rank 0
-- bench.py's grouped step, with the staging to the host that a gloo group needs in place of the on-device RCCL gather.
Rank 0 compares what arrived, row by row, with its OWN single-process forward of the whole batch: bit-identical, in row order.
What is gathered in the reference's terms: standalone.py:3075-3092.  (until now this path had run with
the real encoder on a one-rank group only.)"""

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lengths, out_path):
    sys.path.insert(0, str(ROOT))
    import bench
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.packing import pack_rows
    from open_provence_amd.sharding import agree_on_kernel_set, collective_audit
    from open_provence_amd.synthetic import named_dims, refinit_state_dict, synth_pair_batch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims = named_dims("xsmall", num_layers=3)
        state = refinit_state_dict(dims, seed=7)
        rows_all = synth_pair_batch(dims, len(lengths), lengths, seed=1234)
        enc = HipEncoder(dims, device="cuda:0")
        enc.load_state_dict(state)  # calibrates: every rank for itself, from the same weights
        enc.audit_collective = True
        agreed = agree_on_kernel_set(enc, None)
        verdict = collective_audit(enc, rows_all[:8], None)  # the SAME rows on every rank, verdicts combined
        assert enc.effective_policy()["kernel_set"] == agreed and not enc.audit_pending

        plan, rows, halves = bench.exchange_plan(rows_all, world, rank, dims.num_labels, 2)
        collected = []
        launched = []
        for part, (part_rows, part_plan, rows_j) in enumerate(halves):  # both launch sequences enqueued, then collected
            ids_np, cu_np, max_len = pack_rows(part_rows)
            dev = enc.device
            ids, cu = torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev)
            keep = torch.empty(int(cu_np[-1]), dtype=torch.float32, device=dev)
            seg = torch.tensor(bench.fragment_range_list(cu_np), dtype=torch.int32, device=dev).reshape(-1, 2)
            enc.pipeline_stream(part).wait_stream(torch.cuda.current_stream())
            _prune, rank_logits = enc.forward_packed_on(part, ids, cu, cu_np, max_len, keep_prob=keep)
            with torch.cuda.stream(enc.pipeline_stream(part)):
                means = enc.segment_means(keep, seg)
            launched.append((part_plan, rows_j, means, rank_logits, part))
        for part_plan, rows_j, means, rank_logits, part in launched:
            enc.pipeline_stream(part).synchronize()
            got = part_plan.gather(means.cpu(), rank_logits.cpu(), dst=0)
            if rank == 0:
                vals, rank_all = got
                flat, pcu = vals.reshape(-1), part_plan.cu
                collected.append({"rows": rows_j, "means": [flat[pcu[i]: pcu[i + 1]].clone() for i in range(len(rows_j))], "rank": rank_all.clone()})
            else:
                assert got is None
        if rank == 0:
            # the whole batch in ONE process, one launch sequence
            ids_np, cu_np, max_len = pack_rows(rows_all)
            dev = enc.device
            keep = torch.empty(int(cu_np[-1]), dtype=torch.float32, device=dev)
            _p, rank_one = enc.forward_packed(torch.from_numpy(ids_np).to(dev), torch.from_numpy(cu_np).to(dev), cu_np, max_len, keep_prob=keep)
            seg = torch.tensor(bench.fragment_range_list(cu_np), dtype=torch.int32, device=dev).reshape(-1, 2)
            means_one = enc.segment_means(keep, seg).cpu()
            torch.cuda.synchronize()
            counts = [(n + bench.FRAGMENT_TOKENS - 1) // bench.FRAGMENT_TOKENS for n in lengths]
            starts = np.concatenate([[0], np.cumsum(counts)])
            torch.save({"collected": collected, "rank_one": rank_one.cpu(), "kernel_set": agreed, "verdict": verdict,
                        "means_one": [means_one[starts[i]: starts[i + 1]].clone() for i in range(len(lengths))]}, out_path)
        enc.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_gather_what_one_process_computes(tmp_path):
    lengths = [512, 40, 33, 300, 64, 65, 128, 450, 97, 512, 200, 31, 480, 256, 70, 140]
    out_path = str(tmp_path / "two_ranks.pt")
    mp.spawn(_worker, args=(2, _free_port(), lengths, out_path), nprocs=2, join=True)
    got = torch.load(out_path, weights_only=False)
    assert got["kernel_set"] == "f16"
    seen = []
    for part in got["collected"]:
        for k, row in enumerate(part["rows"]):
            assert torch.equal(part["means"][k], got["means_one"][row]), row  # bit-identical, at the row's own position
            assert torch.equal(part["rank"][k], got["rank_one"][row]), row
            seen.append(row)
    assert sorted(seen) == list(range(len(lengths)))
