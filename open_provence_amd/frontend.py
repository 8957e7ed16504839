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
"""

from __future__ import annotations

import atexit
import os
import socket
from typing import Any, Callable

__all__ = ["ProcessFrontEnd"]


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
