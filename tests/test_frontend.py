"""ProcessFrontEnd: ``process()`` with its host stages spread over worker processes equals the plain call (CPU, gloo)."""

import pytest

from helpers import frontend_stub_model, period_splitter


def _request():
    words = "the tower is tall boats carry fish and salt to north city harbour many years ago it was new".split()
    contexts = [
        " ".join(" ".join(words[(i * 5 + s * 3 + k) % len(words)] for k in range(4 + (i + s) % 5)).capitalize() + "." for s in range(1 + i % 6))
        for i in range(70)
    ]
    return dict(question="which boats carry salt?", context=contexts, sentence_splitter=period_splitter, show_progress=False,
                return_sentence_metrics=True, return_sentence_texts=True, batch_size=8, threshold=0.4)


@pytest.mark.timeout(300)
def test_front_end_equals_the_plain_call_and_survives_a_bad_request():
    from open_provence_amd.frontend import ProcessFrontEnd

    plain_model = frontend_stub_model()
    want = plain_model.process(**_request())
    want_top = plain_model.process(reorder=True, top_k=9, **_request())
    with ProcessFrontEnd(frontend_stub_model, workers=2) as front:
        got = front.process(**_request())
        got_top = front.process(reorder=True, top_k=9, **_request())
        # a request that fails on every rank (unknown language for the built-in splitter table) raises here and leaves
        # the workers serving
        with pytest.raises((ValueError, RuntimeError)):
            bad = _request()
            bad.pop("sentence_splitter")
            front.process(language="xx", **bad)
        again = front.process(**_request())
    for key in want:
        if key in ("timing", "performance_trace"):
            continue
        assert got[key] == want[key], key
        assert again[key] == want[key], key
        assert got_top[key] == want_top[key], key
    assert len(got_top["pruned_context"]) == 9


@pytest.mark.timeout(300)
def test_host_front_end_equals_the_plain_call_and_survives_a_bad_request():
    """mode "host": the workers are host-stage replicas without a model; every forward batch goes through a pipe to the
    caller's process (here a stub forward: per-token keep-probabilities come back, the replicas take the means)."""

    from open_provence_amd.frontend import HostFrontEnd

    plain_model = frontend_stub_model()
    want = plain_model.process(**_request())
    want_top = plain_model.process(reorder=True, top_k=9, **_request())
    with HostFrontEnd(frontend_stub_model(), workers=3) as front:
        got = front.process(**_request())
        got_top = front.process(reorder=True, top_k=9, **_request())
        with pytest.raises((ValueError, RuntimeError)):
            bad = _request()
            bad.pop("sentence_splitter")
            front.process(language="xx", **bad)
        again = front.process(**_request())
        few = front.process(question="which boats?", context=["Boats carry salt. The tower is tall."], sentence_splitter=period_splitter,
                            show_progress=False, return_sentence_texts=True, threshold=0.4)  # fewer jobs than replicas
    for key in want:
        if key in ("timing", "performance_trace"):
            continue
        assert got[key] == want[key], key
        assert again[key] == want[key], key
        assert got_top[key] == want_top[key], key
    assert len(got_top["pruned_context"]) == 9
    want_few = plain_model.process(question="which boats?", context=["Boats carry salt. The tower is tall."], sentence_splitter=period_splitter,
                                   show_progress=False, return_sentence_texts=True, threshold=0.4)
    assert few["pruned_context"] == want_few["pruned_context"] and few["kept_sentences"] == want_few["kept_sentences"]
    assert front.model._dist is None  # the owner model is a plain model again


@pytest.mark.timeout(300)
def test_host_front_end_titles_and_several_queries():
    """The owner normalises the request and assigns the jobs once; a replica sees only its own texts -- explicit titles,
    first-line titles and the aligned / nested input shapes must come out as in the plain call."""

    from open_provence_amd.frontend import HostFrontEnd

    base = _request()
    contexts = base["context"][:12]
    common = {k: v for k, v in base.items() if k not in ("question", "context")}
    cases = [
        dict(question=base["question"], context=contexts, title=[f"Title {i}" for i in range(len(contexts))], **common),
        dict(question=base["question"], context=["Heading %d\n%s" % (i, c) for i, c in enumerate(contexts)], first_line_as_title=True, **common),
        dict(question=["which boats?", "how tall is the tower?"], context=[contexts[:5], contexts[5:9]], **common),
        dict(question=["which boats?", "how tall is the tower?"], context=[contexts[0], contexts[1]], **common),
        dict(question="which boats?", context=contexts[3], **common),
        dict(question=base["question"], context=contexts, title="One title for all", **common),
        dict(question=base["question"], context=contexts, title=None, always_select_title=True, **common),
        dict(question=["which boats?", "how tall is the tower?"], context=[contexts[:5], contexts[5:9]], title=["Boats", "Towers"], **common),
    ]
    plain_model = frontend_stub_model()
    with HostFrontEnd(frontend_stub_model(), workers=3) as front:
        for case in cases:
            want, got = plain_model.process(**case), front.process(**case)
            for key in want:
                if key not in ("timing", "performance_trace"):
                    assert got[key] == want[key], (key, list(case)[:3])


def test_host_front_end_refuses_an_unpicklable_tokenizer_and_accepts_a_factory():
    from helpers import build_wordpiece_tokenizer, host_only_model, golden_stub_forward, wordpiece_tokenizer_for_workers
    from open_provence_amd.frontend import HostFrontEnd

    model = host_only_model(tokenizer=build_wordpiece_tokenizer(True), max_length=96, forward=golden_stub_forward)
    with pytest.raises(TypeError, match="tokenizer_factory"):
        HostFrontEnd(model, workers=1)  # (the helper's tokenizer class is local to a function)
    want = model.process(**_request())
    with HostFrontEnd(model, workers=2, tokenizer_factory=wordpiece_tokenizer_for_workers) as front:
        got = front.process(**_request())
    for key in want:
        if key not in ("timing", "performance_trace"):
            assert got[key] == want[key], key


@pytest.mark.timeout(300)
def test_host_front_end_with_a_dead_worker_raises_instead_of_hanging():
    from open_provence_amd.frontend import HostFrontEnd

    front = HostFrontEnd(frontend_stub_model(), workers=2)
    try:
        assert front.process(**_request())["pruned_context"]
        front._procs[1].terminate()
        front._procs[1].join(timeout=30)
        for _ in range(2):  # the request in which the loss is noticed, and the ones after it
            with pytest.raises(RuntimeError, match="worker process has gone"):
                front.process(**_request())
    finally:
        front.close()


@pytest.mark.timeout(300)
def test_process_routes_large_requests_through_replicas_when_the_environment_asks(monkeypatch):
    """OPEN_PROVENCE_HOST_REPLICAS=N: ``model.process()`` itself keeps a HostFrontEnd and uses it for large requests; small
    ones and calls whose arguments cannot be pickled run in-process; the results are the plain call's."""

    plain_model = frontend_stub_model()
    want = plain_model.process(**_request())
    model = frontend_stub_model()
    monkeypatch.setenv("OPEN_PROVENCE_HOST_REPLICAS", "2")
    try:
        got = model.process(**_request())
        front = model.__dict__.get("_host_front_end")
        assert front is not None and front.world == 2 and front.last_trace["rows"] > 0  # the replicas' forwards went through here
        # through the replicas `timing` used to read preprocess = inference = 0 with the whole call under
        # postprocess -- the fields the reference's harness publishes (scripts/eval_datasets.py:225, 359-365).  They now come
        # from the owner's time line of the request: non-zero, and together no more than the call
        timing = got["timing"]
        assert timing["preprocess_seconds"] > 0 and timing["inference_seconds"] > 0 and timing["postprocess_seconds"] >= 0, timing
        assert timing["preprocess_seconds"] + timing["inference_seconds"] + timing["postprocess_seconds"] <= timing["total_seconds"] + 1e-6, timing
        small = dict(_request(), context=_request()["context"][:3])
        assert model.process(**small)["pruned_context"] == plain_model.process(**small)["pruned_context"]
        local = dict(_request(), sentence_splitter=lambda text: period_splitter(text))  # cannot be pickled
        assert model.process(**local)["pruned_context"] == want["pruned_context"]
        # a tokenizer that cannot be pickled: the variable is ignored (one warning), the call runs in-process
        from helpers import build_wordpiece_tokenizer, golden_stub_forward, host_only_model

        odd = host_only_model(tokenizer=build_wordpiece_tokenizer(True), max_length=96, forward=golden_stub_forward)
        assert odd.process(**_request())["pruned_context"] and odd.__dict__.get("_front_end_unavailable") is True
    finally:
        if model.__dict__.get("_host_front_end") is not None:
            model.__dict__["_host_front_end"].close()
    for key in want:
        if key not in ("timing", "performance_trace"):
            assert got[key] == want[key], key
