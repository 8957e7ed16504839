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


def test_shard_plan_round_trip_vectorised():
    from open_provence_amd.sharding import ShardPlan

    rng = np.random.default_rng(3)
    lengths = [int(n) for n in rng.integers(0, 50, size=23)]
    for world in (1, 2, 5):
        for width, nl in ((1, 1), (2, 3)):
            plan = ShardPlan(lengths, world, width=width, num_labels=nl)
            total = sum(lengths)
            values = torch.arange(total * width, dtype=torch.float32).reshape(total, width)
            ranks = torch.arange(len(lengths) * nl, dtype=torch.float32).reshape(len(lengths), nl) + 0.5
            cu = np.concatenate([[0], np.cumsum(lengths)])
            bucket = []
            for r in range(world):
                mine = plan.local_rows(r)
                v = torch.cat([values[cu[i] : cu[i + 1]] for i in mine]) if mine else torch.zeros((0, width))
                bucket.append(plan.pack(r, v, ranks[mine] if mine else torch.zeros((0, nl))))
            tok, rk = plan.unpack(torch.cat(bucket))
            assert torch.equal(tok, values) and torch.equal(rk, ranks)


def _process_worker(rank, world, port, out_path, shard):
    """process() with the golden stub forward on every rank of a gloo group: rank 0's result must equal the
    single-process result exactly; the other ranks return None.  ``shard``: "jobs" (every rank prepares, runs and
    post-processes its own contexts, one gather_object of the results) or "rows" (forward batches divided only)."""

    import json
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from helpers import GOLDEN_DIR, CharTokenizer, golden_stub_forward, host_only_model, period_splitter

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        meta = json.loads((GOLDEN_DIR / "g3_process_stub.json").read_text(encoding="utf-8"))
        model = host_only_model(tokenizer=CharTokenizer(), max_length=meta["max_length"], forward=golden_stub_forward)
        model.attach_process_group(None, dst=0, shard=shard)
        results = []
        for case in meta["cases"]:
            res = model.process(question=case["question"], context=case["context"], sentence_splitter=period_splitter,
                                show_progress=False, return_sentence_metrics=True, return_sentence_texts=True, batch_size=4,
                                **case["kwargs"])
            if rank == 0:
                res.pop("timing")
                res.pop("performance_trace")
                results.append(res)
            else:
                assert res is None
        if rank == 0:
            torch.save(results, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shard,world", [("jobs", 2), ("jobs", 3), ("rows", 2)])
def test_process_sharded_over_gloo_ranks_matches_golden(tmp_path, shard, world):
    import json

    from helpers import GOLDEN_DIR, assert_process_result_matches

    out_path = str(tmp_path / "proc.pt")
    mp.spawn(_process_worker, args=(world, _free_port(), out_path, shard), nprocs=world, join=True)
    results = torch.load(out_path, weights_only=False)
    meta = json.loads((GOLDEN_DIR / "g3_process_stub.json").read_text(encoding="utf-8"))
    assert len(results) == len(meta["cases"])
    for res, case in zip(results, meta["cases"]):
        assert_process_result_matches(res, case["expected"], prob_tol=1e-6, score_tol=1e-6)


def _split_worker(rank, world, port, lengths, out_path):
    from open_provence_amd.sharding import ShardPlan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = ShardPlan(lengths, world, width=1, num_labels=1)
        got = {}
        for j, (plan_j, rows_j) in enumerate(plan.split(2)):  # the order bench.py issues them in: part 0, then part 1
            mine = [rows_j[k] for k in plan_j.local_rows(rank)]
            assert set(mine) <= set(plan.local_rows(rank))
            values = torch.cat([torch.arange(lengths[i], dtype=torch.float32) + 1000.0 * i for i in mine]) if mine else torch.zeros(0)
            ranks = torch.tensor([[float(i)] for i in mine], dtype=torch.float32).reshape(len(mine), 1)
            out = plan_j.gather(values, ranks, dst=0)
            if rank == 0:
                got[j] = (rows_j, out[0].reshape(-1), out[1].reshape(-1))
            else:
                assert out is None
        if rank == 0:
            torch.save(got, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_plans_gather_each_half_in_row_order(tmp_path, world):
    """ShardPlan.split: every rank's rows cut in two, one plan per half (what a rank running two launch sequences
    gathers with) -- the halves partition the rows, each rank's share of a half is a slice of its own rows, and each
    half's gather returns its rows' values in ascending row order."""

    rng = np.random.default_rng(5)
    lengths = rng.integers(1, 30, size=23).tolist()
    out_path = str(tmp_path / "split.pt")
    mp.spawn(_split_worker, args=(world, _free_port(), lengths, out_path), nprocs=world, join=True)
    got = torch.load(out_path)
    assert sorted(i for j in got for i in got[j][0]) == list(range(23))
    for j in got:
        rows_j, tok, rk = got[j]
        want = torch.cat([torch.arange(lengths[i], dtype=torch.float32) + 1000.0 * i for i in rows_j])
        assert torch.equal(tok, want) and torch.equal(rk, torch.tensor([float(i) for i in rows_j]))


def _subgroup_worker(rank, world, port, lengths, out_path):
    """World of 3, group = global ranks (1, 2): the gather's root is GROUP rank 0 = GLOBAL rank 1."""

    from open_provence_amd.sharding import ShardPlan, gather_row_outputs

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        group = dist.new_group(ranks=[1, 2])
        if rank in (1, 2):
            me = dist.get_rank(group)
            plan = ShardPlan(lengths, 2, width=1, num_labels=1)
            rows = plan.local_rows(me)
            values = torch.cat([torch.full((lengths[i],), float(i)) for i in rows]) if rows else torch.zeros(0)
            logits = torch.tensor([[10.0 * i] for i in rows], dtype=torch.float32).reshape(len(rows), 1)
            out = plan.gather(values, logits, dst=0, group=group)
            assert (out is not None) == (me == 0)
            # explicit (non-default) shards through the legacy entry point
            custom = [[i for i in range(len(lengths)) if i % 2 == 0], [i for i in range(len(lengths)) if i % 2 == 1]]
            prune = torch.cat([torch.full((lengths[i], 2), float(i)) for i in custom[me]])
            rk = torch.tensor([[float(i)] for i in custom[me]], dtype=torch.float32)
            legacy = gather_row_outputs(prune, rk, custom[me], lengths, custom, dst=0, group=group)
            if me == 0:
                torch.save({"tok": out[0], "rank": out[1], "legacy_rows": legacy[0], "legacy_rank": legacy[1]}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gather_on_a_real_sub_group_translates_the_root_rank(tmp_path):
    lengths = [5, 3, 8, 2, 7]
    out_path = str(tmp_path / "sub.pt")
    mp.spawn(_subgroup_worker, args=(3, _free_port(), lengths, out_path), nprocs=3, join=True)
    got = torch.load(out_path)
    expect_tok = torch.cat([torch.full((n,), float(i)) for i, n in enumerate(lengths)]).reshape(-1, 1)
    assert torch.equal(got["tok"], expect_tok)
    assert torch.equal(got["rank"], torch.tensor([[10.0 * i] for i in range(len(lengths))]))
    assert [t.shape[0] for t in got["legacy_rows"]] == lengths
    assert all(torch.all(t == float(i)) for i, t in enumerate(got["legacy_rows"]))
    assert torch.equal(got["legacy_rank"], torch.tensor([[float(i)] for i in range(len(lengths))]))


def _collect_worker(rank, world, port, out_path):
    """The group form of the pipelined collect (modeling._collect_rows_sharded) on fabricated launch handles: each
    rank holds the fragment means + ranking logits of ITS rows, rank 0 must get every row's values in row order."""

    import sys
    from pathlib import Path
    from types import SimpleNamespace

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from helpers import CharTokenizer, host_only_model

    from open_provence_amd.sharding import ShardPlan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = host_only_model(tokenizer=CharTokenizer())
        model.dims = SimpleNamespace(num_labels=2)
        model.attach_process_group(None, dst=0)
        lengths = [40, 7, 512, 130, 64, 300, 9]
        counts = [3, 1, 11, 4, 2, 7, 1]  # fragments per row
        plan = ShardPlan(lengths, world, width=1, num_labels=2)
        mine = plan.local_rows(rank)
        for with_segments in (True, False):
            per_row = counts if with_segments else lengths
            vals = np.concatenate([np.arange(per_row[i], dtype=np.float32) + 1000.0 * i for i in mine]) if mine else np.zeros(0, np.float32)
            rk = np.asarray([[float(i), -float(i)] for i in mine], dtype=np.float32).reshape(-1)
            handle = {"event": None, "pool": {"keep_np": vals, "rank_np": rk}, "total": int(sum(lengths[i] for i in mine)),
                      "rows": len(mine), "cu": None, "seg_counts": [counts[i] for i in mine] if with_segments else None, "alive": None,
                      "shard": {"mine": mine, "n_rows_all": len(lengths), "lengths": lengths, "shards": plan.shards,
                                "counts_all": counts if with_segments else None}}
            rank_all, rows_out = model._collect_rows(handle)
            assert rank_all.shape == (len(lengths), 2) and len(rows_out) == len(lengths)
            if rank == 0:
                assert torch.equal(rank_all, torch.tensor([[float(i), -float(i)] for i in range(len(lengths))]))
                for i, got in enumerate(rows_out):
                    want = (np.arange(per_row[i], dtype=np.float32) + 1000.0 * i)
                    assert isinstance(got, list) == with_segments
                    assert np.array_equal(np.asarray(got, dtype=np.float32), want), (with_segments, i)
        if rank == 0:
            torch.save({"ok": True}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_collect_gathers_fragment_means_in_row_order(tmp_path, world):
    out_path = str(tmp_path / "collect.pt")
    mp.spawn(_collect_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    assert torch.load(out_path)["ok"]


def test_assign_jobs_is_deterministic_complete_and_balanced():
    """Job-level sharding of process(): every (query, context) has exactly one owner, the same on every rank, and the
    owners' character loads differ by at most one context."""

    from open_provence_amd.pipeline import assign_jobs

    rng = np.random.default_rng(3)
    contexts = [["x" * int(n) for n in rng.integers(1, 4000, size=37)], [["ab" * 50, "c" * 10]], [], ["y" * 5]]
    for world in (1, 2, 3, 8):
        owner = assign_jobs(contexts, world)
        assert owner == assign_jobs(contexts, world)
        assert [len(o) for o in owner] == [len(c) for c in contexts]
        assert all(0 <= r < world for per in owner for r in per)
        loads = [0] * world
        for per_q, per_o in zip(contexts, owner):
            for entry, r in zip(per_q, per_o):
                loads[r] += (sum(len(s) for s in entry) if isinstance(entry, list) else len(entry)) + 64
        assert max(loads) - min(loads) <= 4000 + 64
    assert assign_jobs([["a", "b"]], 1) == [[0, 0]]


def _many_contexts_worker(rank, world, port, out_path):
    """A 90-context request (several contexts per rank, reorder + top_k on the merged result): job-sharded process() on
    every rank, rank 0 also computes the plain single-process result first; both are stored for comparison."""

    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from helpers import CharTokenizer, golden_stub_forward, host_only_model, period_splitter

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        words = "the tower is tall boats carry fish and salt to north city harbour many years ago it was new".split()
        contexts = [
            " ".join(" ".join(words[(i * 5 + s * 3 + k) % len(words)] for k in range(4 + (i + s) % 5)).capitalize() + "." for s in range(1 + i % 6))
            for i in range(90)
        ]
        kwargs = dict(question="which boats carry salt?", context=contexts, sentence_splitter=period_splitter, show_progress=False,
                      return_sentence_metrics=True, return_sentence_texts=True, batch_size=8, threshold=0.4, reorder=True, top_k=25)
        model = host_only_model(tokenizer=CharTokenizer(), max_length=96, forward=golden_stub_forward)
        plain = model.process(**kwargs) if rank == 0 else None
        model.attach_process_group(None, dst=0, shard="jobs")
        got = model.process(**kwargs)
        if rank == 0:
            for res in (plain, got):
                res.pop("timing")
                res.pop("performance_trace")
            torch.save({"plain": plain, "sharded": got}, out_path)
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_job_sharded_process_of_many_contexts_equals_the_plain_call(tmp_path, world):
    out_path = str(tmp_path / "many.pt")
    mp.spawn(_many_contexts_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    both = torch.load(out_path, weights_only=False)
    assert both["plain"].keys() == both["sharded"].keys()
    for key in both["plain"]:
        assert both["plain"][key] == both["sharded"][key], key
    assert len(both["sharded"]["pruned_context"]) == 25  # top_k applied on the merged, re-ordered result


def _failing_rank_worker(rank, world, port, out_path):
    """Job-sharded process() where ONE rank's share fails (its splitter raises on a marked context): every rank must
    come back with an exception -- the failing one with its own, the others with a report of who failed -- instead of
    the healthy ranks blocking in the final gather forever."""

    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from helpers import CharTokenizer, golden_stub_forward, host_only_model, period_splitter
    from open_provence_amd.pipeline import assign_jobs

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        contexts = [f"Sentence one of context {i}. Sentence two of it is here. And a third one." for i in range(12)]
        owner = assign_jobs([contexts], world)[0]
        victim = next(i for i, r in enumerate(owner) if r == world - 1)  # a context the LAST rank owns
        contexts[victim] = "POISON " + contexts[victim]
        owner = assign_jobs([contexts], world)[0]
        bad_rank = owner[victim]

        def splitter(text):
            if text.startswith("POISON"):
                raise ValueError("splitter failed on purpose")
            return period_splitter(text)

        model = host_only_model(tokenizer=CharTokenizer(), max_length=96, forward=golden_stub_forward)
        model.attach_process_group(None, dst=0, shard="jobs")
        outcome = "returned"
        try:
            model.process("which sentence?", contexts, sentence_splitter=splitter, show_progress=False)
        except ValueError as exc:
            outcome = f"own:{exc}"
        except RuntimeError as exc:
            outcome = f"peer:{exc}"
        torch.save({"outcome": outcome, "bad_rank": bad_rank}, f"{out_path}.{rank}")
        # the group is still usable afterwards: a clean call goes through
        clean = model.process("which sentence?", [c for c in contexts if not c.startswith("POISON")], sentence_splitter=splitter, show_progress=False)
        assert (clean is not None) == (rank == 0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_a_failing_rank_does_not_strand_the_others_in_the_gather(tmp_path, world):
    out_path = str(tmp_path / "fail.pt")
    mp.spawn(_failing_rank_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    reports = [torch.load(f"{out_path}.{r}", weights_only=False) for r in range(world)]
    bad = reports[0]["bad_rank"]
    for r, rep in enumerate(reports):
        if r == bad:
            assert rep["outcome"].startswith("own:") and "on purpose" in rep["outcome"], rep
        else:
            assert rep["outcome"].startswith("peer:") and f"{bad}: ValueError" in rep["outcome"], rep


# ---- bench.py --gpus N: the exchange a timed step performs (exchange_plan -> fragment means -> ShardPlan.gather) -----------
def _bench_exchange_worker(rank, world, port, rows_all, n_pipes, out_path):
    """What bench.py's step does between the forward and the end of the step, with a stand-in encoder (keep-probability of a
    token = a function of its id; ranking logit of a row = a function of its ids): per-fragment means of THIS rank's rows,
    one gather per launch sequence.  Rank 0 stores what arrived, in the order the plan restores."""

    import bench

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan, rows, halves = bench.exchange_plan(rows_all, world, rank, 1, n_pipes)
        parts = [(part_rows, part_plan, rows_j) for part_rows, part_plan, rows_j in halves] if n_pipes == 2 else [(rows, plan, list(range(len(rows_all))))]
        collected = []
        for part_rows, part_plan, rows_j in parts:
            keep = np.array([(t % 97) / 97.0 for r in part_rows for t in r], dtype=np.float32)
            cu = np.zeros(len(part_rows) + 1, dtype=np.int64)
            for i, r in enumerate(part_rows):
                cu[i + 1] = cu[i] + len(r)
            means = torch.tensor([float(keep[a:b].mean()) for a, b in bench.fragment_range_list(cu)], dtype=torch.float32)
            rank_logits = torch.tensor([[float(sum(r) % 1013)] for r in part_rows], dtype=torch.float32).reshape(len(part_rows), 1)
            got = part_plan.gather(means, rank_logits, dst=0)
            if rank == 0:
                vals, rank_all = got
                flat, pcu = vals.reshape(-1), part_plan.cu
                collected.append({"rows": rows_j, "means": [flat[pcu[i] : pcu[i + 1]].clone() for i in range(len(rows_j))], "rank": rank_all.clone()})
            else:
                assert got is None
        if rank == 0:
            torch.save(collected, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pipes", [(2, 1), (2, 2), (3, 2)])
def test_bench_exchange_lands_in_row_order_for_unequal_rows(tmp_path, world, n_pipes):
    """bench.py:exchange_plan (rows by token count -> plan over fragment counts -> split(2)) had only ever run on a one-rank
    group; here world-size 2 / 3 on gloo with rows of very different lengths: every row's fragment means and ranking logit
    arrive on rank 0 at the row's own position.  Reference for what is gathered: standalone.py:3075-3092."""

    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import bench

    rng = np.random.default_rng(11)
    lengths = [512, 33, 1, 700, 64, 65, 31, 300, 128, 2, 450, 97, 512]
    rows_all = [rng.integers(1, 5000, size=n).tolist() for n in lengths]
    out_path = str(tmp_path / "exchange.pt")
    mp.spawn(_bench_exchange_worker, args=(world, _free_port(), rows_all, n_pipes, out_path), nprocs=world, join=True)
    collected = torch.load(out_path)
    seen = []
    for part in collected:
        for k, row in enumerate(part["rows"]):
            ids = np.array(rows_all[row])
            keep = ((ids % 97) / 97.0).astype(np.float32)
            want = [float(keep[a : a + bench.FRAGMENT_TOKENS].mean()) for a in range(0, len(ids), bench.FRAGMENT_TOKENS)]
            assert part["means"][k].tolist() == pytest.approx(want, abs=0, rel=0) or np.array_equal(part["means"][k].numpy(), np.array(want, dtype=np.float32)), row
            assert float(part["rank"][k, 0]) == float(sum(rows_all[row]) % 1013), row
            seen.append(row)
    assert sorted(seen) == list(range(len(rows_all)))  # every row exactly once across the launch sequences


class _FakeEncoder:
    """Stand-in for HipEncoder's kernel-set / audit surface (test infrastructure): what sharding.agree_on_kernel_set and
    sharding.collective_audit talk to."""

    def __init__(self, chosen, audit_verdict, default="f16-f8-w"):
        self.kernel_set, self.default, self.verdict, self.audit_pending = chosen, default, audit_verdict, chosen != default
        self.audited_rows, self.reverted = None, None

    def effective_policy(self):
        return {"kernel_set": self.kernel_set}

    def audit_rows(self, rows):
        self.audited_rows, self.audit_pending = [list(r) for r in rows], False
        return self.verdict

    def revert_to_default(self, reason):
        self.kernel_set, self.reverted, self.audit_pending = self.default, reason, False
        return self.kernel_set


def _one_arithmetic_worker(rank, world, port, case, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from open_provence_amd.sharding import agree_on_kernel_set, collective_audit

        rows = [[1, 2, 3, 4] * 20, [5, 6] * 40]
        if case == "audit fails on one rank":
            enc = _FakeEncoder("f16", audit_verdict=(rank != world - 1))
        elif case == "load-time choices differ":
            enc = _FakeEncoder("f16" if rank == 0 else "bf16", audit_verdict=True)
        else:
            enc = _FakeEncoder("f16", audit_verdict=True)
        agreed = agree_on_kernel_set(enc, None)
        verdict = collective_audit(enc, rows, None)
        again = collective_audit(enc, rows, None)  # nothing pending any more: no collective, no audit
        torch.save({"agreed": agreed, "verdict": verdict, "again": again, "set": enc.effective_policy()["kernel_set"],
                    "audited": enc.audited_rows, "reverted": enc.reverted}, f"{out_path}.{rank}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["all pass", "audit fails on one rank", "load-time choices differ"])
def test_every_rank_of_a_job_runs_the_same_kernel_set(tmp_path, case):
    """each rank used to audit its own shard, and a rank that failed alone went back to the
    default selection alone -- outputs depending on how the batch was cut, one rank 1.8 x slower inside a synchronous
    gather.  Now the ranks compare their load-time choices once and audit the SAME rows with the verdicts combined by MIN:
    one failing rank (or one differing choice) sends EVERY rank to the default selection."""

    world, port, out = 2, _free_port(), str(tmp_path / "out")
    mp.spawn(_one_arithmetic_worker, args=(world, port, case, out), nprocs=world, join=True)
    got = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    assert got[0]["set"] == got[1]["set"]
    assert all(g["again"] is None for g in got)
    if case == "all pass":
        assert all(g["set"] == "f16" and g["verdict"] is True and g["reverted"] is None for g in got)
        assert got[0]["audited"] == got[1]["audited"] and got[0]["audited"] is not None  # the same rows on every rank
    elif case == "audit fails on one rank":
        assert all(g["set"] == "f16-f8-w" and g["verdict"] is False and "audit failed" in g["reverted"] for g in got)
    else:
        assert all(g["agreed"] == "f16-f8-w" and g["set"] == "f16-f8-w" and "different kernel sets" in g["reverted"] for g in got)
        assert all(g["audited"] is None for g in got)  # (reverted at attach time: nothing left to audit)
