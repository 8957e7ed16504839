import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_library():
    from open_provence_amd import build_ext, _lib

    build_ext.build()
    return _lib.load_library()


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a HIP device: without one (the CPU build container) they are skipped rather
    than failed, so a plain ``pytest tests`` is green there.  On a GPU box nothing is skipped -- a missing
    library still fails loudly inside the tests (no silent fallback)."""

    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
