"""Host front-end of ``process()`` on several PROCESSES, one GPU (or one per process).

``process()`` spends most of a call with short contexts on the host -- sentence splitting, tokenizing, fragment decoding,
block assembly, post-processing -- and a Python interpreter runs those on one core: with a Hugging Face fast tokenizer the
GPU is busy ~10 % of a call.  The reference hands the stage to ``DataLoader`` worker processes
(``standalone.py:3589``, ``:2478-2519``).  Here the unit that scales is the whole per-context pipeline: ``process()`` over
a process group already shards the (query, context) JOBS (``attach_process_group(shard="jobs")``: every rank splits,
tokenizes, assembles, runs and post-processes only the contexts it owns; one ``gather_object`` at the end).  This module
keeps N - 1 such ranks alive as worker processes behind ONE caller:

    front = ProcessFrontEnd(build_model, workers=7)          # build_model() -> OpenProvenceModel, picklable, no args
    result = front.process(question, contexts, threshold=0.1)   # same arguments, same result as model.process(...)
    front.close()

* the caller is rank 0 of a ``gloo`` group and holds a model of its own; every worker builds its model once
  (``build_model`` runs in the worker: same checkpoint, same device or another one) and then serves requests;
* a request is one ``broadcast_object_list`` of the call's arguments; every rank runs ``model.process(...)`` on its
  share; the per-context results come back through the gather of the job-sharded path; the caller applies ``reorder`` /
  ``top_k`` exactly as ``process()`` does -- so the result equals the plain call's, field for field (the forward of a row
  does not depend on its batch companions: bit-identical per-row outputs, ``tests/test_gpu_model.py``);
* workers are started with the ``spawn`` method (no fork next to a live HIP runtime) and stopped by ``close()`` (or
  at interpreter exit); a worker that fails reports through the error path of the gather -- every rank raises.

Several ranks on ONE GPU is deliberate: the weights of the path's models are MBs, the forward of a 256-context share is
a few ms, and the GPU is otherwise idle while the hosts tokenize.

``mode="host"`` (``HostFrontEnd``): the workers hold NO model and never touch the GPU.  Ten processes with a HIP context
each time-slice one GPU with forwards of ~100 contexts that fill a fifth of it: at 1024 contexts the call has a floor of
~30 ms of mutual waiting whatever the number of workers (profiles/r04_process_e2e.txt: 9 workers 19.3 k contexts/s, 12:
16.6 k, 23: 13.2 k).  Here a worker is a *host-stage replica* (``OpenProvenceModel._host_stage_replica``: tokenizer +
request settings, no encoder): it splits, tokenizes, assembles and post-processes its share of the jobs and sends every
forward batch -- packed ids, row offsets, fragment ranges: a few numpy buffers -- through a pipe to the caller's process,
which owns the GPU, takes no jobs itself, MERGES the batches that are waiting into one launch, and returns ranking logits
+ per-fragment means (4 bytes per fragment).  Same job assignment, same per-row arithmetic (a row's outputs do not depend
on its batch companions), so the result equals the plain call's.  Measured (profiles/r04_process_e2e.txt; the GPU box's
container has a CPU quota of 16 cores -- cpu.max 1600000 100000 -- so every figure with more than 16 busy processes is
throttled): 1024 contexts, WordPiece tokenizer: 24-27 k contexts/s with 31 replicas (ProcessFrontEnd, 9 workers: 19.8 k;
one process: 4.85 k), 4096 contexts: 30.6 k, 256: 18.3 k.  ``HostFrontEnd.last_trace`` holds the owner's time line of the
last request (launch sizes and times, the slowest replica's stamps).
"""

from __future__ import annotations

import atexit
import os
import socket
import threading
from time import perf_counter, process_time
from typing import Any, Callable

__all__ = ["ProcessFrontEnd", "HostFrontEnd", "default_host_workers", "usable_cores"]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def _serve(rank: int, world: int, port: int, build_model: Callable[[], Any]) -> None:
    """Worker main: join the group, build the model, serve requests until the caller says stop."""

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = build_model()
        model.attach_process_group(None, dst=0, shard="jobs")
        dist.barrier()  # the caller's constructor returns once every worker is ready
        while True:
            box: list[Any] = [None]
            dist.broadcast_object_list(box, src=0)
            request = box[0]
            if request is None:
                break
            args, kwargs = request
            try:
                model.process(*args, **kwargs)  # this rank's share; the result goes to rank 0 through the gather
            except Exception:
                # reported to the caller (and every peer) by the job-sharded path's error marker; keep serving
                continue
    finally:
        dist.destroy_process_group()


class ProcessFrontEnd:
    """``workers`` extra processes that share the host stages of every ``process()`` call with the caller's own model."""

    def __init__(self, build_model: Callable[[], Any], workers: int = 3, *, model: Any | None = None) -> None:
        import torch.distributed as dist
        import torch.multiprocessing as mp

        if workers < 1:
            raise ValueError("workers must be >= 1 (use model.process() directly otherwise)")
        if dist.is_initialized():
            raise RuntimeError("ProcessFrontEnd creates its own default process group; one is already initialised")
        self.world = int(workers) + 1
        port = _free_port()
        ctx = mp.get_context("spawn")
        self._procs = [ctx.Process(target=_serve, args=(r, self.world, port, build_model), daemon=True) for r in range(1, self.world)]
        for proc in self._procs:
            proc.start()
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=0, world_size=self.world)
        self.model = model if model is not None else build_model()
        self.model.attach_process_group(None, dst=0, shard="jobs")
        dist.barrier()
        self._open = True
        atexit.register(self.close)

    def process(self, *args: Any, **kwargs: Any):
        """``OpenProvenceModel.process`` with the jobs spread over the caller and its workers."""

        import torch.distributed as dist

        if not self._open:
            raise RuntimeError("the front-end has been closed")
        dist.broadcast_object_list([(args, kwargs)], src=0)
        return self.model.process(*args, **kwargs)

    def close(self) -> None:
        import torch.distributed as dist

        if not getattr(self, "_open", False):
            return
        self._open = False
        try:
            dist.broadcast_object_list([None], src=0)
        finally:
            self.model.attach_process_group(enabled=False)
            for proc in self._procs:
                proc.join(timeout=30)
                if proc.is_alive():
                    proc.terminate()
            if dist.is_initialized():
                dist.destroy_process_group()

    def __enter__(self) -> "ProcessFrontEnd":
        return self

    def __exit__(self, *_exc: Any) -> None:
        self.close()


# ------------------------------------------------------------------------------------------------------------------
# mode "host": replicas without a GPU, one owner of the forward
# ------------------------------------------------------------------------------------------------------------------
class _ReplicaLink:
    """Replica side of the pipe: the remote forward (``submit`` / ``result``) and the delivery of the replica's part."""

    def __init__(self, conn: Any, vocab_size: int | None, num_labels: int) -> None:
        self.conn = conn
        self.vocab_size = vocab_size
        self.num_labels = int(num_labels)
        self.request = -1
        self.delivered = True
        self.stamps: dict[str, float] = {}

    # -- forward ---------------------------------------------------------------------------------------------------
    def submit(self, rows: list[list[int]], segments: list[list[tuple[int, int]]] | None) -> dict[str, Any]:
        from .modeling import OpenProvenceModel

        ids_np, cu_np, max_len, seg_flat, seg_counts = OpenProvenceModel._pack_launch(rows, segments)
        if self.vocab_size is not None and ids_np.size and (int(ids_np.min()) < 0 or int(ids_np.max()) >= self.vocab_size):
            raise IndexError(f"token id out of range for the embedding table: ids span [{int(ids_np.min())}, {int(ids_np.max())}], "
                             f"vocab_size is {self.vocab_size}")  # (what HipEncoder.check_ids raises in the plain call)
        self.conn.send(("rows", self.request, (ids_np, cu_np, int(max_len), seg_flat, seg_counts)))
        self.stamps.setdefault("first_submit", perf_counter())
        return {"remote": self, "rows": len(rows), "cu": cu_np, "seg_counts": seg_counts}

    def result(self, handle: dict[str, Any]):
        import torch

        kind, request, payload = self.conn.recv()  # replies come back in the order of the submissions
        self.stamps["last_result"] = perf_counter()
        if kind == "error" or request != self.request:
            raise RuntimeError(f"the GPU owner could not run a forward batch: {payload}")
        rank_np, values, reduced = payload
        n_rows = handle["rows"]
        rank = torch.from_numpy(rank_np).reshape(n_rows, self.num_labels)
        if reduced:
            means, out, pos = values.tolist(), [], 0  # float32 -> Python float, exact (as OpenProvenceModel._collect_rows)
            for c in handle["seg_counts"]:
                out.append(means[pos : pos + c])
                pos += c
            return rank, out
        cu = handle["cu"]
        return rank, [values[cu[i] : cu[i + 1]] for i in range(n_rows)]

    # -- this replica's part of the result (OpenProvenceModel._gather_job_results) --------------------------------------
    def exchange(self, _model: Any, mine: dict) -> list:
        # (perf_counter is the system-wide monotonic clock: the owner can place these stamps on its own time line)
        self.conn.send(("result", self.request, mine, dict(self.stamps, part_sent=perf_counter(),
                                                           cpu_seconds=process_time() - self.stamps.get("cpu0", process_time()))))
        self.delivered = True
        return []


def _serve_host_stages(rank: int, world: int, conn: Any, spec: dict) -> None:
    """Worker main of mode "host": a host-stage replica that serves requests until the owner says stop."""

    import pickle
    from multiprocessing import resource_tracker, shared_memory

    from .modeling import OpenProvenceModel

    # (a replica only ATTACHES to the owner's request blocks: its resource tracker must not unlink them at exit)
    _register = resource_tracker.register
    resource_tracker.register = lambda name, rtype: None if rtype == "shared_memory" else _register(name, rtype)
    link = _ReplicaLink(conn, spec.get("vocab_size"), spec.get("num_labels", 1))
    model = OpenProvenceModel._host_stage_replica(spec, link)
    model._dist = {"group": None, "dst": world, "rank": rank, "world": world, "force": True, "shard": "jobs", "local_only": False,
                   "transport": link}
    conn.send(("ready", -1, None))
    while True:
        message = conn.recv()
        if message is None:
            break
        if isinstance(message[0], str):
            continue  # the answer to a forward batch of a request this replica has already given up on
        link.request, shm_name, size, mine = message
        link.delivered = False
        link.stamps = {"request_received": perf_counter(), "cpu0": process_time()}
        try:
            block = shared_memory.SharedMemory(name=shm_name)
            try:
                kwargs, queries = pickle.loads(bytes(block.buf[:size]))
            finally:
                block.close()
            contexts, own_titles = mine  # per query: the texts this replica owns (the owner knows where they belong)
            if own_titles is not None:
                kwargs = dict(kwargs, title=own_titles)
            model._dist["prenormalized"] = {"queries": queries, "contexts": contexts, "structure": "nested",
                                            "owner": [[rank] * len(per_query) for per_query in contexts]}
            model.process(None, None, **kwargs)
        except Exception as exc:  # noqa: BLE001 - reported to the owner, which raises for the caller; keep serving
            if not link.delivered:
                link.exchange(model, {"__error__": f"{type(exc).__name__}: {exc}"})


class _OwnerHub:
    """Owner side: hands requests out, serves the replicas' forward batches, collects their parts."""

    def __init__(self, conns: list[Any]) -> None:
        self.conns = list(conns)
        self.index = {id(c): i for i, c in enumerate(self.conns)}
        self.request = 0
        self.parts: list[Any] = [None] * len(self.conns)
        self.waiting: list[Any] = []

    def _split_request(self, model: Any, args: tuple, kwargs: dict):
        """Normalise the request and assign its jobs ONCE, here: -> (what every replica shares: the keyword arguments
        without question / context, the queries; per replica its own COMPACT request: per query the texts it owns -- and
        their titles, when titles were given per context --; per replica and query the positions of those contexts in the
        caller's request).  A replica then runs an ordinary small ``process()``: nothing in it grows with the size of the
        whole request (normalising, assigning and post-processing placeholders for all N contexts in every replica was
        ~2-4 ms per 1024 contexts and replica -- a sixth of all the CPU time of a call with 31 replicas)."""

        import inspect

        from . import pipeline as pl

        bound = inspect.signature(model.process).bind(*args, **kwargs)
        call = dict(bound.arguments)
        question, context = call.pop("question"), call.pop("context")
        queries, contexts, _structure = model._normalize_inputs(question, context)
        first_line = bool(call.get("first_line_as_title", False))
        resolved, titles = model._resolve_titles(queries, contexts, call.get("title", "first_sentence"), first_line_as_title=first_line)
        owner = pl.assign_jobs(resolved, len(self.conns))  # (on the texts process() assigns on: after the title pass)
        per_context_titles = (not first_line) and any(isinstance(t, list) for t in titles)
        n = len(self.conns)
        positions = [[[] for _ in queries] for _ in range(n)]
        texts = [[[] for _ in queries] for _ in range(n)]
        for q_idx, per_query in enumerate(contexts):
            for c_idx, entry in enumerate(per_query):
                rank = owner[q_idx][c_idx]
                positions[rank][q_idx].append(c_idx)
                texts[rank][q_idx].append(entry)
        shares = []
        for rank in range(n):
            own_titles = None
            if per_context_titles:  # prepare_titles' per-context form: a list of titles per query
                own_titles = [[titles[q][c] for c in positions[rank][q]] if isinstance(titles[q], list) else titles[q]
                              for q in range(len(queries))]
            shares.append((texts[rank], own_titles))
        self.positions = positions
        return call, queries, shares

    def begin(self, model: Any, args: tuple, kwargs: dict) -> None:
        """One request to every replica: the shared part (settings, queries) is pickled ONCE into a shared-memory block,
        each replica's own compact request goes down its pipe."""

        import pickle
        from multiprocessing import shared_memory

        from time import perf_counter

        t0 = perf_counter()
        self._cpu0 = process_time()
        self.request += 1
        self.parts = [None] * len(self.conns)
        self.waiting = list(self.conns)
        self.trace = {"request_bytes": 0, "begin_seconds": 0.0, "serve_seconds": 0.0, "launches": 0, "batches": 0, "rows": 0,
                      "first_batch_seconds": None, "last_part_seconds": None, "t0": t0}
        kwargs, queries, shares = self._split_request(model, args, kwargs)
        blob = pickle.dumps((kwargs, queries), protocol=pickle.HIGHEST_PROTOCOL)
        self.release()
        self._shm = shared_memory.SharedMemory(create=True, size=max(len(blob), 1))
        self._shm.buf[: len(blob)] = blob
        for conn, share in zip(self.conns, shares):
            try:
                conn.send((self.request, self._shm.name, len(blob), share))
            except (OSError, ValueError):  # the replica's process is gone: its jobs cannot be done -- the request fails, loudly
                self.parts[self.index[id(conn)]] = {"__error__": "the worker process has gone (create a new HostFrontEnd)"}
                self.waiting.remove(conn)
        self.trace["request_bytes"] = len(blob)
        self.trace["begin_seconds"] = perf_counter() - t0

    def release(self) -> None:
        shm = getattr(self, "_shm", None)
        if shm is not None:
            self._shm = None
            shm.close()
            shm.unlink()

    def exchange(self, model: Any, _mine: dict) -> list:
        """Called from the owner's ``process()`` where the collective of the group path would be (it owns no job, so it
        arrives here at once): serve until every replica has delivered.  Returns the parts by replica rank."""

        self.serve(model)
        return list(self.parts)

    def serve(self, model: Any) -> None:
        """Until every replica has delivered its part: receive forward batches, run them, send the outputs back.

        Launch policy: ONE launch in flight; whatever arrives while it runs is accumulated and becomes the next launch the
        moment it completes.  A forward of 100 contexts is latency-bound (~2 ms whatever its size below ~500 contexts), so
        ten launches of 100 take three times as long as three launches that grow with the backlog -- and the GPU still
        starts on the first batch that arrives."""

        from multiprocessing.connection import wait
        from time import perf_counter

        t_serve = perf_counter()
        trace = self.trace
        inflight: tuple | None = None
        backlog: list = []
        while self.waiting or inflight is not None or backlog:
            if inflight is not None and self._finished(inflight):
                t1 = perf_counter()
                self._reply(model, inflight)
                trace["reply_seconds"] = trace.get("reply_seconds", 0.0) + perf_counter() - t1
                trace.setdefault("reply_at", []).append(round((perf_counter() - trace["t0"]) * 1e3, 1))
                inflight = None
            if inflight is None and backlog:
                if trace["first_batch_seconds"] is None:
                    trace["first_batch_seconds"] = perf_counter() - trace["t0"]
                trace["launches"] += 1
                trace["batches"] += len(backlog)
                trace["rows"] += sum(len(payload[1]) - 1 for _conn, payload in backlog)
                batch, backlog = backlog, []
                trace.setdefault("launch_at", []).append((round((perf_counter() - trace["t0"]) * 1e3, 1), sum(len(p[1]) - 1 for _c, p in batch)))
                t1 = perf_counter()
                try:
                    inflight = self._launch(model, batch)
                    trace["launch_seconds"] = trace.get("launch_seconds", 0.0) + perf_counter() - t1
                except Exception as exc:  # noqa: BLE001 - the replicas raise it inside their process() and report back
                    for conn, _payload in batch:
                        conn.send(("error", self.request, f"{type(exc).__name__}: {exc}"))
            if not self.waiting:
                if inflight is not None:  # nothing more can arrive: wait for the launch
                    self._reply(model, inflight)
                    inflight = None
                continue
            # idle: block on the pipes; a launch in flight: look at them for a moment, then at the launch again
            t1 = perf_counter()
            ready = wait(self.waiting, timeout=None if inflight is None else 0.0002)
            trace["wait_seconds" if inflight is None else "poll_seconds"] = trace.get("wait_seconds" if inflight is None else "poll_seconds", 0.0) + perf_counter() - t1
            t1 = perf_counter()
            for conn in ready:
                stamps = None
                try:
                    kind, request, payload, *rest = conn.recv()
                    stamps = rest[0] if rest else None
                except (EOFError, OSError):
                    kind, request, payload = "result", self.request, {"__error__": "the worker process has gone"}
                if kind == "rows" and request != self.request:  # (cannot happen: a request ends when every part is in)
                    conn.send(("error", request, "stale forward batch"))
                elif kind == "rows":
                    backlog.append((conn, payload))
                elif kind == "result" and request == self.request:
                    who = self.index[id(conn)]
                    if isinstance(payload, dict) and "__error__" not in payload:  # a replica counts ITS contexts: back to the caller's
                        payload = {(q, self.positions[who][q][c]): values for (q, c), values in payload.items()}
                    self.parts[who] = payload
                    self.waiting.remove(conn)
                    trace["last_part_seconds"] = perf_counter() - trace["t0"]
                    if stamps:  # CPU time of all replicas; the slowest replica's time line, relative to the start of the request
                        trace["replica_cpu_seconds"] = trace.get("replica_cpu_seconds", 0.0) + stamps.pop("cpu_seconds", 0.0)
                        stamps.pop("cpu0", None)
                        rel = {k: v - trace["t0"] for k, v in stamps.items()}
                        if rel.get("part_sent", 0.0) >= trace.get("replica", {}).get("part_sent", 0.0):
                            trace["replica"] = rel
            trace["recv_seconds"] = trace.get("recv_seconds", 0.0) + perf_counter() - t1
        trace["serve_seconds"] += perf_counter() - t_serve

    @staticmethod
    def _finished(launched: tuple) -> bool:
        if launched[0] != "handle":
            return True  # (replaced forward: computed synchronously)
        event = launched[2].get("event")
        return event is None or bool(event.query())

    def _launch(self, model: Any, batch: list) -> tuple:
        """ONE launch for all the batches that are waiting: their packed rows are concatenated (a row's outputs do not
        depend on its companions), the reply is cut back per replica.  payload = (ids, cu, max_len, range positions |
        None, ranges per row | None) as ``OpenProvenceModel._pack_launch`` makes it."""

        import numpy as np
        import torch

        native = model._forward_is_native()
        with_ranges = all(payload[4] is not None for _conn, payload in batch)
        if len(batch) > 1 and not (native and with_ranges):
            # (a replaced forward -- the CPU tests --, or payloads without ranges: one launch each, replies in order)
            return ("many", [self._launch(model, [item]) for item in batch])
        cu_parts, seg_parts, seg_counts, requests = [np.zeros(1, dtype=np.int32)], [], [], []
        tokens = 0
        for conn, (_ids, cu_np, _max_len, seg_flat, counts) in batch:
            n_tok = int(cu_np[-1])
            cu_parts.append(np.asarray(cu_np[1:], dtype=np.int32) + np.int32(tokens))
            if with_ranges:
                seg_parts.append(np.asarray(seg_flat, dtype=np.int32) + np.int32(tokens))
                seg_counts.extend(counts)
            requests.append((conn, len(cu_np) - 1, sum(counts) if with_ranges else n_tok, n_tok))
            tokens += n_tok
        ids = np.concatenate([payload[0] for _conn, payload in batch])
        cu = np.concatenate(cu_parts)
        max_len = max(int(payload[2]) for _conn, payload in batch)
        if native:
            handle = model._enqueue_packed(ids, cu, max_len, np.concatenate(seg_parts) if with_ranges else None,
                                           seg_counts if with_ranges else None)
            return ("handle", requests, handle)
        # a model whose forward was replaced (tests): the padded protocol, per-token keep-probabilities
        rows = [ids[cu[i] : cu[i + 1]].tolist() for i in range(len(cu) - 1)]
        rank, keeps = model._predict_rows_local(rows, None)
        values = np.concatenate([np.asarray(k, dtype=np.float32)[: len(r)] for k, r in zip(keeps, rows)]) if rows else np.zeros(0, np.float32)
        requests = [(conn, n_rows, n_tok, n_tok) for conn, n_rows, _vals, n_tok in requests]
        return ("done", requests, (rank.to("cpu", torch.float32).numpy().reshape(len(rows), -1), values, False))

    def _reply(self, model: Any, launched: tuple) -> None:
        if launched[0] == "many":
            for one in launched[1]:
                self._reply(model, one)
            return
        kind, requests, what = launched
        try:
            rank, values, reduced = model._collect_packed(what) if kind == "handle" else what
        except Exception as exc:  # noqa: BLE001
            for conn, _rows, _vals, _tok in requests:
                conn.send(("error", self.request, f"{type(exc).__name__}: {exc}"))
            return
        row0 = val0 = 0
        for conn, n_rows, n_vals, n_tok in requests:
            if not reduced:  # the launch fell back to per-token values (_enqueue_packed dropped the ranges): cut by tokens
                n_vals = n_tok
            conn.send(("out", self.request, (rank[row0 : row0 + n_rows].copy(), values[val0 : val0 + n_vals].copy(), reduced)))
            row0 += n_rows
            val0 += n_vals


def usable_cores() -> float:
    """Cores this process may keep busy: the container's CPU quota (cgroup v2 ``cpu.max`` / v1 cfs quota) if there is one,
    else its affinity mask.  ``os.cpu_count()`` reports the HOST's CPUs: on the GPU boxes of this project 256 under a quota
    of 16."""

    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return max(1.0, float(quota) / float(period))
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        if quota > 0:
            return max(1.0, quota / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))
    except (OSError, ValueError):
        pass
    try:
        return float(len(os.sched_getaffinity(0)))
    except AttributeError:  # pragma: no cover - not Linux
        return float(os.cpu_count() or 1)


def default_host_workers() -> int:
    """Replicas ``HostFrontEnd`` starts when the caller does not say: twice the usable cores minus the owner, at most 31
    (measured under a quota of 16 cores: 15 replicas 19-21 k contexts/s, 23: 22-25 k, 31: 24-27 k at 1024 contexts -- a
    replica spends part of a request waiting for the owner's answers, so a modest oversubscription pays)."""

    return int(max(1, min(31, 2 * usable_cores() - 1)))


_MAIN_HIDE_LOCK = threading.Lock()


class _main_module_hidden:
    """While active (and ``enabled``), ``multiprocessing``'s spawn finds nothing to re-import as the children's main module:
    ``spawn.get_preparation_data`` reads ``__main__.__spec__.name`` / ``__main__.__file__`` when a process is STARTED."""

    def __init__(self, enabled: bool) -> None:
        self.enabled = enabled
        self.saved: tuple | None = None

    def __enter__(self) -> None:
        if not self.enabled:
            return
        import sys

        _MAIN_HIDE_LOCK.acquire()
        main = sys.modules.get("__main__")
        missing = object()
        self.saved = (main, getattr(main, "__spec__", missing), getattr(main, "__file__", missing), missing)
        if main is not None:
            main.__spec__ = None
            if hasattr(main, "__file__"):
                del main.__file__

    def __exit__(self, *_exc: Any) -> None:
        if not self.enabled:
            return
        main, spec, file, missing = self.saved
        if main is not None:
            if spec is not missing:
                main.__spec__ = spec
            if file is not missing:
                main.__file__ = file
        _MAIN_HIDE_LOCK.release()


class HostFrontEnd:
    """``workers`` host-stage replicas (no GPU, no model) behind ``model``, whose process owns every forward.

        front = HostFrontEnd(model)                                  # workers=None: default_host_workers()
        result = front.process(question, contexts, threshold=0.1)   # = model.process(...)
        front.close()

    Arguments of ``process`` must be picklable (a ``sentence_splitter`` callable has to be importable by name), and so must
    the model's tokenizer (Hugging Face tokenizers are) -- or pass ``tokenizer_factory``, a callable importable by name
    that builds the same tokenizer inside each replica."""

    def __init__(self, model: Any, workers: int | None = None, *, tokenizer_factory: Callable[[], Any] | None = None,
                 import_main: bool = True) -> None:
        """``import_main=False``: the workers do NOT import the caller's ``__main__`` module (multiprocessing's spawn does
        by default, so that objects defined there unpickle -- and a script without an ``if __name__ == "__main__"`` guard runs
        again in every worker).  What ``OpenProvenceModel.process`` uses when it starts a front-end by itself; callables
        defined in ``__main__`` cannot be sent to such workers."""

        import pickle

        import torch.multiprocessing as mp

        if workers is None:
            workers = default_host_workers()
        if workers < 1:
            raise ValueError("workers must be >= 1 (use model.process() directly otherwise)")
        if getattr(model, "_dist", None):
            raise RuntimeError("the model already has a process group attached")
        self.model = model
        self.world = int(workers)
        ctx = mp.get_context("spawn")  # (no fork next to a live HIP runtime)
        spec = model._host_stage_spec()
        if tokenizer_factory is not None:
            spec["tokenizer"], spec["tokenizer_factory"] = None, tokenizer_factory
        try:
            pickle.dumps(spec)
        except Exception as exc:
            raise TypeError("HostFrontEnd sends the model's tokenizer to its worker processes and it cannot be pickled "
                            f"({type(exc).__name__}: {exc}); pass tokenizer_factory=<a module-level function that builds it>") from exc
        self._procs, conns = [], []
        try:
            with _main_module_hidden(not import_main):
                for rank in range(self.world):
                    mine, theirs = ctx.Pipe(duplex=True)
                    proc = ctx.Process(target=_serve_host_stages, args=(rank, self.world, theirs, spec), daemon=True)
                    proc.start()
                    theirs.close()
                    self._procs.append(proc)
                    conns.append(mine)
            for conn in conns:
                kind, _request, _payload = conn.recv()  # "ready": the replica has its tokenizer
                if kind != "ready":
                    raise RuntimeError("a host-stage worker did not start")
        except BaseException:
            # a later start() or handshake failed: the replicas that DID start must not outlive this constructor (the caller
            # never gets an object to close; modeling.process() catches the exception and carries on in-process)
            for conn in conns:
                try:
                    conn.close()
                except OSError:
                    pass
            for proc in self._procs:
                if proc.is_alive():
                    proc.terminate()
            for proc in self._procs:
                proc.join(timeout=5)
            self._procs = []
            raise
        self._hub = _OwnerHub(conns)
        self._open = True
        self._lock = threading.Lock()
        atexit.register(self.close)

    def process(self, *args: Any, **kwargs: Any):
        """``OpenProvenceModel.process`` with the host stages on the replicas and every forward on this process's GPU.
        One request at a time (calls from several threads are serialised here); while a request runs the model is the
        owner of a job-sharded call -- do not call ``model.process`` directly from another thread meanwhile."""

        if not self._open:
            raise RuntimeError("the front-end has been closed")
        with self._lock:
            return self._process_locked(args, kwargs)

    def _process_locked(self, args: tuple, kwargs: dict):
        hub, model = self._hub, self.model
        hub.begin(model, args, kwargs)
        model._dist = {"group": None, "dst": self.world, "rank": self.world, "world": self.world, "force": True, "shard": "jobs",
                       "local_only": False, "transport": hub}
        try:
            result = model.process(*args, **kwargs)
            self.last_trace = {k: v for k, v in hub.trace.items() if k != "t0"}  # where the owner's time went (seconds)
            self.last_trace["owner_cpu_seconds"] = process_time() - hub._cpu0
            return result
        finally:
            model._dist = None
            if hub.waiting:  # the owner left early (its own exception): the replicas still get served and heard
                hub.serve(model)
            hub.release()

    def close(self) -> None:
        if not getattr(self, "_open", False):
            return
        self._open = False
        self._hub.release()
        for conn in self._hub.conns:
            try:
                conn.send(None)
            except (OSError, BrokenPipeError):
                pass
        for proc in self._procs:
            proc.join(timeout=30)
            if proc.is_alive():
                proc.terminate()

    def __enter__(self) -> "HostFrontEnd":
        return self

    def __exit__(self, *_exc: Any) -> None:
        self.close()
