"""The C-ABI library must load on a CPU-only box and export every symbol include/open_provence_hip.h
declares; without a GPU its entry points must fail loudly (no fallback), never compute."""

import ctypes
import re
from pathlib import Path

import pytest
import torch

from open_provence_amd import _lib

HEADER = Path(__file__).resolve().parents[1] / "include" / "open_provence_hip.h"


def declared_functions() -> list[str]:
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(op_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(hip_library):
    for name in declared_functions():
        assert hasattr(hip_library, name), name
    assert hip_library.op_abi_version() == _lib.OP_ABI_VERSION


def test_config_struct_size_is_checked(hip_library):
    cfg = _lib.OpConfig()
    cfg.struct_bytes = 12  # wrong on purpose
    handle = ctypes.c_void_p()
    code = hip_library.op_create(ctypes.byref(cfg), ctypes.byref(handle))
    assert code == -1 and not handle.value
    assert "struct_bytes" in _lib.last_error(hip_library, None)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure(hip_library):
    from open_provence_amd.engine import HipEncoder
    from open_provence_amd.synthetic import named_dims

    with pytest.raises(_lib.HipLibraryError, match="no CPU fallback"):
        HipEncoder(named_dims("xsmall", vocab_size=1024))
    cfg = _lib.OpConfig()
    cfg.struct_bytes = ctypes.sizeof(_lib.OpConfig)
    cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size = 256, 128, 128
    cfg.num_layers, cfg.num_heads, cfg.num_labels = 2, 2, 1
    cfg.local_attention, cfg.max_position_embeddings = 128, 512
    cfg.norm_eps, cfg.global_rope_theta, cfg.local_rope_theta = 1e-5, 160000.0, 10000.0
    handle = ctypes.c_void_p()
    code = hip_library.op_create(ctypes.byref(cfg), ctypes.byref(handle))
    assert code == -3 and not handle.value
    assert "no CPU fallback" in _lib.last_error(hip_library, None)


def test_unsupported_head_dim_is_rejected_before_touching_the_device(hip_library):
    cfg = _lib.OpConfig()
    cfg.struct_bytes = ctypes.sizeof(_lib.OpConfig)
    cfg.vocab_size, cfg.hidden_size, cfg.intermediate_size = 256, 64, 96
    cfg.num_layers, cfg.num_heads, cfg.num_labels = 2, 4, 1
    cfg.local_attention, cfg.max_position_embeddings = 16, 512
    handle = ctypes.c_void_p()
    code = hip_library.op_create(ctypes.byref(cfg), ctypes.byref(handle))
    assert code == -2
    assert "head_dim" in _lib.last_error(hip_library, None)


def test_product_kernel_headers_carry_no_ablation_switches(tmp_path):
    """The whole-layer kernel's ablation hooks (OPK_ABL_*, OPK_QKV_*, OPK_PLAIN_*: builds with wrong results that price one
    component of a loop) live in microbench/experiments/rowgemm_ablation_hooks.patch, not in the shipped headers; the
    patch must still apply to them (scripts/instrumented_csrc.sh is what the microbenchmarks compile against)."""

    import shutil
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    check = subprocess.run([sys.executable, str(root / "scripts" / "strip_switches.py"), "--check"], capture_output=True, text=True)
    assert check.returncode == 0, check.stdout
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    dst = tmp_path / "open_provence_amd"
    dst.mkdir()
    shutil.copytree(root / "open_provence_amd" / "csrc", dst / "csrc")
    with open(root / "microbench" / "experiments" / "rowgemm_ablation_hooks.patch") as fh:
        applied = subprocess.run(["patch", "-p1", "-s"], stdin=fh, cwd=tmp_path, capture_output=True, text=True)
    assert applied.returncode == 0, applied.stdout + applied.stderr
    assert "OPK_ABL_NO_DMA" in (dst / "csrc" / "opk_rowgemm_mlp_loop.inc").read_text()  # (a section of RowGemmBlock::mlp_phase)


def test_named_dims_overrides_are_checked():
    """An override that named_dims does not know must not be dropped silently (``num_layers=3`` once ran the full-depth model);
    ``num_layers`` is accepted as the EncoderDims name of ``num_hidden_layers``."""

    from open_provence_amd.synthetic import named_dims

    assert named_dims("base").num_layers == 19
    assert named_dims("base", num_layers=3).num_layers == 3 == named_dims("base", num_hidden_layers=3).num_layers
    assert named_dims("xsmall", vocab_size=2048).vocab_size == 2048
    with pytest.raises(TypeError):
        named_dims("base", layers=3)


def test_calibration_tolerance_argument_forms(monkeypatch):
    """``calibrate=1`` (an int, not ``True``) used to become a tolerance of 1.0 -- any candidate passed; ``np.bool_``
    likewise.  Booleans are recognised by type, ``1`` means "on", and a tolerance at or above the path's own bar is refused."""

    import numpy as np
    import pytest

    from open_provence_amd.engine import DEFAULT_CALIBRATION_TOLERANCE, resolve_calibration_tolerance

    monkeypatch.delenv("OPEN_PROVENCE_CALIBRATE", raising=False)
    for on in (True, None, 1, np.True_, np.int64(1)):
        assert resolve_calibration_tolerance(on) == DEFAULT_CALIBRATION_TOLERANCE, on
    for off in (False, 0, 0.0, np.False_):
        assert resolve_calibration_tolerance(off) == 0.0, off
    assert resolve_calibration_tolerance(2e-4) == 2e-4
    for bad in (1.0, 1e-3, 5, float("nan")):
        with pytest.raises(ValueError):
            resolve_calibration_tolerance(bad)
    monkeypatch.setenv("OPEN_PROVENCE_CALIBRATE", "3e-5")
    assert resolve_calibration_tolerance(None) == 3e-5
    monkeypatch.setenv("OPEN_PROVENCE_CALIBRATE", "0.5")
    with pytest.raises(ValueError):
        resolve_calibration_tolerance(True)
    monkeypatch.setenv("OPEN_PROVENCE_CALIBRATE", "off")
    assert resolve_calibration_tolerance(None) == 0.0
