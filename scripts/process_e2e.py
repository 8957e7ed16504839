#!/usr/bin/env python
"""GPU-box probe: end-to-end ``process()`` (host pipeline + HIP forward) on a synthetic 1-query x N-context
request with the char tokenizer of the tests; prints the reference-style timing breakdown and, with
--profile, the top of a cProfile run.  Usage: scripts/process_e2e.py [--contexts 256] [--chars 470] [--profile]"""
import argparse
import cProfile
import io
import json
import os
import pstats
import resource
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np
import torch

from helpers import CharTokenizer, period_splitter


def make_request(n_contexts: int, chars: int, seed: int = 5):
    rng = np.random.default_rng(seed)
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu"]
    contexts = []
    for _ in range(n_contexts):
        parts, total = [], 0
        while total < chars:
            n = int(rng.integers(5, 12))
            sent = " ".join(words[int(i)] for i in rng.integers(0, len(words), n)) + ". "
            parts.append(sent)
            total += len(sent)
        contexts.append("".join(parts)[:chars].rstrip() + ".")
    return "which greek letters appear here", contexts


def build_e2e_tokenizer():
    """Tokenizer factory (module level: HostFrontEnd's replicas rebuild it by name); kind from the environment."""

    if os.environ.get("E2E_TOKENIZER", "char") == "wordpiece":
        from helpers import build_wordpiece_tokenizer

        return build_wordpiece_tokenizer(True, legacy_methods=os.environ.get("E2E_STOCK_TOKENIZER") != "1")
    return CharTokenizer()


def build_e2e_model():
    """Model factory (module level: ProcessFrontEnd's worker processes rebuild it by name); tokenizer from the environment."""

    from open_provence_amd.config import OpenProvenceConfig
    from open_provence_amd.modeling import OpenProvenceModel
    from open_provence_amd.synthetic import named_dims, synth_state_dict

    dims = named_dims("xsmall")
    cfg = OpenProvenceConfig(base_model_config=dims.to_base_model_config(), tokenizer_name_or_path="x",
                             pruning_config={"hidden_size": dims.hidden_size}, max_length=512)
    if os.environ.get("E2E_TOKENIZER", "char") == "wordpiece":
        from helpers import build_wordpiece_tokenizer

        tok = build_wordpiece_tokenizer(True, legacy_methods=os.environ.get("E2E_STOCK_TOKENIZER") != "1")
    else:
        tok = CharTokenizer()
    model = OpenProvenceModel(cfg, device="cuda", tokenizer=tok, state_dict=synth_state_dict(dims, 7))
    model.tokenizer.model_max_length = 512
    return model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--front-end", type=int, default=0,
                    help="N > 0: open_provence_amd.frontend.ProcessFrontEnd with N worker processes (all on cuda:0) beside the caller")
    ap.add_argument("--host-front-end", type=int, default=0,
                    help="N > 0: open_provence_amd.frontend.HostFrontEnd -- N host-stage replicas WITHOUT a GPU; this process runs every forward")
    ap.add_argument("--contexts", type=int, default=256)
    ap.add_argument("--chars", type=int, default=470)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--preprocess-batch", type=int, default=None, help="explicit preprocess batch (default: the pipeline granule)")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--tokenizer", default="char", choices=["char", "wordpiece"],
                    help="char: the pure-Python tokenizer of the tests; wordpiece: a Hugging Face fast (Rust) tokenizer built offline")
    ap.add_argument("--workers", type=int, default=None, help="preprocess_workers (worker threads of the split / tokenize stage)")
    ap.add_argument("--stock-tokenizer", action="store_true",
                    help="wordpiece: the plain transformers PreTrainedTokenizerFast object (what a checkpoint's tokenizer files load as; "
                    "it pickles, so process() can start host replicas by itself) instead of the test helper's local 4.x-style subclass")
    ap.add_argument("--chars-are-words", action="store_true", help="wordpiece: size the contexts in words (~tokens) instead of characters")
    args = ap.parse_args()

    os.environ["E2E_TOKENIZER"] = args.tokenizer
    if args.stock_tokenizer:
        os.environ["E2E_STOCK_TOKENIZER"] = "1"
    question, contexts = make_request(args.contexts, args.chars)
    front = None
    if args.front_end > 0:
        from open_provence_amd.frontend import ProcessFrontEnd

        front = ProcessFrontEnd(build_e2e_model, workers=args.front_end)
        target = front
    elif args.host_front_end > 0:
        from open_provence_amd.frontend import HostFrontEnd

        front = HostFrontEnd(build_e2e_model(), workers=args.host_front_end, tokenizer_factory=build_e2e_tokenizer)
        target = front
    else:
        target = build_e2e_model()

    def call():
        return target.process(question, contexts, threshold=0.1, batch_size=args.batch_size, sentence_splitter=period_splitter,
                              show_progress=False, preprocess_batch_size=args.preprocess_batch, preprocess_workers=args.workers)

    call()
    torch.cuda.synchronize()
    best = None
    for _ in range(args.reps):
        r0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        out = call()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r1 = resource.getrusage(resource.RUSAGE_SELF)
        usage = {"user_s": round(r1.ru_utime - r0.ru_utime, 4), "sys_s": round(r1.ru_stime - r0.ru_stime, 4),
                 "minor_faults": r1.ru_minflt - r0.ru_minflt, "vol_ctx_switches": r1.ru_nvcsw - r0.ru_nvcsw,
                 "invol_ctx_switches": r1.ru_nivcsw - r0.ru_nivcsw}
        if best is None or dt < best[0]:
            best = (dt, out["timing"], usage)
    dt, timing, usage = best
    print(json.dumps({"contexts": args.contexts, "chars": args.chars, "tokenizer": args.tokenizer, "workers": args.workers, "front_end_processes": args.front_end, "host_replicas": args.host_front_end, "wall_s": dt, "contexts_per_s": args.contexts / dt, "rusage": usage, "owner_trace": getattr(front, "last_trace", None),
                      "timing": {k: round(float(v), 5) for k, v in timing.items()}}))
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        call()
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(35)
        print(buf.getvalue()[:6000])
    if front is not None:
        front.close()


if __name__ == "__main__":
    main()
