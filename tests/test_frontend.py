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
