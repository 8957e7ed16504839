#!/usr/bin/env python
"""Differential fuzz of the HOST pipeline against the real reference (build container only: imports
/root/reference through tests/golden/make_golden.load_reference): random ``process()`` requests -- every input shape,
titles in every accepted form, thresholds, the flags that change sentence selection and scoring, batch sizes -- through
the reference's ``OpenProvenceModel.process`` and through this package's, both with the SAME deterministic replacement
forward (the goldens' stub), so that every difference is a difference in host semantics.  Fields are compared exactly.

    python scripts/reference_diff_fuzz.py [--requests 300] [--seed 0] [--no-specials]
"""
import argparse
import random
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "golden"))

from helpers import CharTokenizer, golden_stub_forward, host_only_model, period_splitter  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-specials", action="store_true", help="a tokenizer that does not emit special tokens (manual-specials path)")
    args = ap.parse_args()
    import make_golden as mg

    emit = not args.no_specials
    ref = mg.load_reference(emit_specials=emit)
    ref_model, _dims = mg.build_model(ref, mg.base_cfg(vocab_size=256, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, local_attention=32), max_length=96, seed=0, weight_seed=None)
    ref_model.forward = golden_stub_forward
    ours = host_only_model(CharTokenizer(emit_specials=emit), max_length=96, forward=golden_stub_forward)

    rng = random.Random(args.seed)
    words = "the tower is tall boats carry fish and salt to north city harbour many years ago it was new".split()

    def sentence() -> str:
        return " ".join(rng.choice(words) for _ in range(rng.randint(1, 9))).capitalize() + rng.choice([".", ".", "!", "?"])

    def ctx() -> str:
        parts = [sentence() for _ in range(rng.randint(0, 9))]
        sep = rng.choice([" ", " ", "\n", "  "])
        text = sep.join(parts)
        if rng.random() < 0.1:
            text = "\n" + text
        if rng.random() < 0.1:
            text += " trailing words without a period"
        return text

    bad = 0
    for trial in range(args.requests):
        shape = rng.choice(["str", "list", "list", "nested", "aligned", "presplit"])
        if shape == "str":
            q, c = "which boats carry salt?", ctx()
        elif shape == "list":
            q, c = "which boats carry salt?", [ctx() for _ in range(rng.randint(1, 9))]
        elif shape == "aligned":
            q = [f"question {i}?" for i in range(rng.randint(2, 4))]
            c = [ctx() for _ in q]
        elif shape == "presplit":
            q, c = "which boats carry salt?", [[sentence() + " " for _ in range(rng.randint(1, 6))] for _ in range(rng.randint(1, 4))]
        else:
            q = [f"question {i}?" for i in range(rng.randint(2, 3))]
            c = [[ctx() for _ in range(rng.randint(1, 5))] for _ in q]
        kw = dict(threshold=rng.choice([0.05, 0.3, 0.5, 0.8]), batch_size=rng.choice([1, 4, 32]),
                  always_select_title=rng.random() < 0.3, use_best_reranker_score=rng.random() < 0.7,
                  zero_score_when_empty=rng.random() < 0.7, strip_sentences=rng.random() < 0.3,
                  respect_sentence_boundaries=rng.random() < 0.3)
        roll = rng.random()
        if roll < 0.15:
            kw["title"] = None
        elif roll < 0.3 and shape in ("list", "presplit"):
            kw["title"] = [f"Title {i}" for i in range(len(c))]
        elif roll < 0.4:
            kw["title"] = "One title"
        elif roll < 0.5 and shape != "presplit":
            kw["first_line_as_title"] = True
            kw.pop("title", None)
        if rng.random() < 0.25 and shape in ("list", "nested"):
            kw["reorder"], kw["top_k"] = True, rng.choice([None, 1, 3])
        call = dict(question=q, context=c, sentence_splitter=period_splitter, show_progress=False, return_sentence_metrics=True,
                    return_sentence_texts=True, **kw)
        outcome = []
        for model in (ref_model, ours):
            try:
                with torch.no_grad():
                    res = dict(model.process(**call))
                res.pop("timing", None)
                res.pop("performance_trace", None)
                outcome.append(mg._jsonable(res))
            except Exception as exc:  # noqa: BLE001 - an error is an outcome too: both sides must agree on its type
                outcome.append(f"{type(exc).__name__}")
        if outcome[0] != outcome[1]:
            bad += 1
            keys = [k for k in outcome[0] if outcome[0][k] != outcome[1].get(k)] if isinstance(outcome[0], dict) and isinstance(outcome[1], dict) else outcome
            print("MISMATCH", trial, shape, {k: v for k, v in kw.items()}, "fields", keys if isinstance(keys, list) else keys, flush=True)
            if bad <= 3 and isinstance(outcome[0], dict) and isinstance(outcome[1], dict):
                for k in keys[:2]:
                    print("   ref :", str(outcome[0][k])[:300])
                    print("   ours:", str(outcome[1][k])[:300])
    # ---- the single-block API: get_raw_predictions_batch / predict_with_thresholds (standalone.py:1742-1890) -------------
    import numpy as np

    api_calls = max(20, args.requests // 4)
    for trial in range(api_calls):
        n = rng.randint(1, 5)
        batch = [[sentence() + " " for _ in range(rng.randint(1, 6))] for _ in range(n)]
        query = [f"question {i}?" for i in range(n)] if rng.random() < 0.4 else "which boats carry salt?"
        bs = rng.choice([None, 1, 2, 8])
        with torch.no_grad():
            want = ref_model.get_raw_predictions_batch(query, batch, batch_size=bs)
            got = ours.get_raw_predictions_batch(query, batch, batch_size=bs)
        ok = len(want) == len(got)
        for w, g in zip(want, got):
            end = max((e for _s, e in w.context_ranges), default=0)  # (past a row's tokens the reference holds what its model
            # makes of pad tokens, this package 0.5: no range addresses those positions)
            ok = ok and w.query == g.query and list(w.contexts) == list(g.contexts) and w.ranking_score == g.ranking_score
            ok = ok and [tuple(r) for r in w.context_ranges] == [tuple(r) for r in g.context_ranges]
            ok = ok and np.array_equal(np.asarray(w.pruning_probs)[:end], np.asarray(g.pruning_probs)[:end])
        thresholds = [0.1, 0.5, 0.9]
        majority = rng.random() < 0.5
        with torch.no_grad():
            pw = ref_model.predict_with_thresholds("which boats?", batch[0], thresholds, use_majority=majority)
            pg = ours.predict_with_thresholds("which boats?", batch[0], thresholds, use_majority=majority)
        ok = ok and pw["predictions"] == pg["predictions"] and pw["ranking_score"] == pg["ranking_score"]
        if not ok:
            bad += 1
            print("MISMATCH single-block API", trial, n, bs, majority, flush=True)
    print(f"requests {args.requests} + {api_calls} single-block API calls, mismatches {bad}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
