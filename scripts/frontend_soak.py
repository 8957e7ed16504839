#!/usr/bin/env python
"""Soak of the front-ends against the plain call: random requests (every input shape, titles, reorder / top_k, batch sizes)
through ``HostFrontEnd`` and through ``model.process`` -- every field must be equal, bit for bit.  On the GPU box with the
real model (default), or ``--stub`` on a CPU box (the replaced forward of the tests).

    python scripts/frontend_soak.py [--requests 150] [--workers 4] [--stub]
"""
import argparse
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "scripts"))

from helpers import frontend_stub_model, period_splitter  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=150)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--stub", action="store_true")
    ap.add_argument("--ranks", action="store_true", help="ProcessFrontEnd (worker ranks with a model each, gloo) instead of HostFrontEnd")
    ap.add_argument("--self", dest="self_check", action="store_true",
                    help="no front-end: the plain call against itself with another batch size, preprocess batch and worker threads")
    args = ap.parse_args()
    from open_provence_amd.frontend import HostFrontEnd

    if args.stub:
        plain, owner = frontend_stub_model(), frontend_stub_model()
    else:
        from process_e2e import build_e2e_model

        plain = owner = build_e2e_model()  # (char tokenizer; the front-end leaves the model a plain model between calls)
    words = "the tower is tall boats carry fish and salt to north city harbour many years ago it was new".split()
    rng = random.Random(11)

    def ctx() -> str:
        return " ".join(" ".join(rng.choice(words) for _ in range(rng.randint(2, 9))).capitalize() + "." for _ in range(rng.randint(1, 8)))

    bad, t0 = 0, time.time()
    import contextlib

    if args.ranks:
        from open_provence_amd.frontend import ProcessFrontEnd

        factory = frontend_stub_model
        if not args.stub:
            from process_e2e import build_e2e_model as factory
        make_front = lambda: ProcessFrontEnd(factory, workers=args.workers, model=owner)  # noqa: E731
    else:
        make_front = lambda: HostFrontEnd(owner, workers=args.workers)  # noqa: E731
    with (contextlib.nullcontext() if args.self_check else make_front()) as front:
        for trial in range(args.requests):
            shape = rng.choice(["list", "nested", "str", "aligned"])
            if shape == "str":
                q, c = "which boats carry salt?", ctx()
            elif shape == "list":
                q, c = "which boats carry salt?", [ctx() for _ in range(rng.randint(1, 60))]
            elif shape == "aligned":
                q = [f"question {i}?" for i in range(rng.randint(2, 4))]
                c = [ctx() for _ in q]
            else:
                q = [f"question {i}?" for i in range(rng.randint(2, 3))]
                c = [[ctx() for _ in range(rng.randint(0, 25))] for _ in q]
            kw = dict(sentence_splitter=period_splitter, show_progress=False, return_sentence_metrics=True, return_sentence_texts=True,
                      batch_size=rng.choice([4, 8, 32]), threshold=rng.choice([0.1, 0.4, 0.6]))
            if rng.random() < 0.3 and shape in ("list", "nested"):
                kw["reorder"], kw["top_k"] = True, rng.choice([None, 3])
            if rng.random() < 0.2 and shape == "list":
                kw["title"] = [f"T{i}" for i in range(len(c))]
            want = plain.process(q, c, **kw)
            if args.self_check:
                other = dict(kw, batch_size=rng.choice([1, 3, 16, 64]), preprocess_batch_size=rng.choice([None, 5, 40]),
                             preprocess_workers=rng.choice([None, 0, 2]))
                got = owner.process(q, c, **other)
            else:
                got = front.process(q, c, **kw)
            for key in want:
                if key not in ("timing", "performance_trace") and want[key] != got[key]:
                    bad += 1
                    print("MISMATCH", trial, shape, key, flush=True)
                    break
    print(f"requests {args.requests} mismatches {bad} seconds {time.time() - t0:.1f}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
