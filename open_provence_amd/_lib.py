"""ctypes binding of ``libopenprovence_hip.so`` (C ABI: ``include/open_provence_hip.h``).

The library is built in-tree by :func:`open_provence_amd.build_ext.build` (hipcc, gfx950).  There is no
fallback of any kind: if the shared object is missing, or there is no HIP device, the product path
raises.
"""

from __future__ import annotations

import ctypes
import os
from pathlib import Path

OP_MAX_LAYERS = 128
OP_ABI_VERSION = 9

OP_OK = 0
OP_ERR_INVALID, OP_ERR_UNSUPPORTED, OP_ERR_HIP, OP_ERR_STATE, OP_ERR_WORKSPACE, OP_ERR_NOMEM = -1, -2, -3, -4, -5, -6
OP_DTYPE_F32, OP_DTYPE_BF16, OP_DTYPE_F16 = 0, 1, 2
OP_PRECISION_BF16X3, OP_PRECISION_BF16, OP_PRECISION_BF16X2, OP_PRECISION_CUSTOM = 0, 1, 2, 3
OP_TERM_LEFT_LO, OP_TERM_RIGHT_LO = 1, 2
# enum op_gemm_family: one term mask per contraction
OP_FAMILIES = ("wqkv", "qk", "pv", "attn_out", "wi", "mlp_out")
OP_FLAG_FORCE_TILED, OP_FLAG_NO_SMALL_BLOCKS, OP_FLAG_ATT_WAVES_4, OP_FLAG_ATT_WAVES_8, OP_FLAG_NO_POLICY_KERNELS = 1, 2, 4, 8, 16
OP_FLAG_NO_LAYER_FUSION = 32
OP_FLAG_LAYER_8X16 = 64
OP_FLAG_LAYER_M32 = 128
OP_FLAG_NO_HEAD_FUSION = 256
OP_FLAG_NO_F8 = 512
OP_FLAG_ATTN_XCD_GROUP = 1024
OP_FLAG_PANEL_F8 = 2048
OP_FLAG_PANEL_F8_WI = 4096
OP_FLAG_NO_LAYER_PAIRS = 8192
OP_POOL_CLS, OP_POOL_MEAN = 0, 1
# enum op_kernel_set (op_effective_policy / op_select_kernel_set / op_calibrate)
OP_KS_AUTO = -1
KERNEL_SET_NAMES = {0: "bf16x3", 1: "bf16-weights", 2: "bf16", 3: "f16-f8", 4: "f16-f8-w", 5: "bf16x3+wi-f16-f8-w",
                    6: "bf16-weights+wi-f16-f8", 7: "f16", 8: "f16+mlp-f16-f8-w", 9: "f16+mlp-f16-f8",
                    10: "f16-f8-w+attn-f16", 11: "f16-f8+attn-f16",
                    -1: "all-terms kernels, cleared lo operands"}
KERNEL_SET_IDS = {name: number for number, name in KERNEL_SET_NAMES.items() if number >= 0}
# kernel sets with an fp16 operand plane: an activation beyond fp16's range comes out as NaN (range guard in engine.py)
FP16_PLANE_SETS = ("f16-f8", "f16-f8-w", "bf16x3+wi-f16-f8-w", "bf16-weights+wi-f16-f8", "f16", "f16+mlp-f16-f8-w", "f16+mlp-f16-f8",
                   "f16-f8-w+attn-f16", "f16-f8+attn-f16")

LIB_NAME = "libopenprovence_hip.so"

# every symbol include/open_provence_hip.h declares (checked by tests/test_abi.py)
EXPORTED_SYMBOLS = (
    "op_abi_version",
    "op_create",
    "op_load_weight",
    "op_weights_ready",
    "op_effective_policy",
    "op_set_compact_operands",
    "op_select_kernel_set",
    "op_mlp_correction_layers",
    "op_select_mlp_correction_layers",
    "op_calibrate",
    "op_workspace_bytes",
    "op_forward_packed",
    "op_segment_means",
    "op_debug_capture_hidden",
    "op_profile_enable",
    "op_profile_read",
    "op_profile_reset",
    "op_profile_kind_name",
    "op_device_count",
    "op_debug_clock_probe",
    "op_destroy",
    "op_last_error",
)


class HipLibraryError(RuntimeError):
    """The HIP library is missing, failed to load, or returned an error code."""


class OpConfig(ctypes.Structure):
    _fields_ = [
        ("struct_bytes", ctypes.c_uint32),
        ("device_id", ctypes.c_int32),
        ("vocab_size", ctypes.c_int32),
        ("hidden_size", ctypes.c_int32),
        ("intermediate_size", ctypes.c_int32),
        ("num_layers", ctypes.c_int32),
        ("num_heads", ctypes.c_int32),
        ("num_labels", ctypes.c_int32),
        ("local_attention", ctypes.c_int32),
        ("max_position_embeddings", ctypes.c_int32),
        ("pooling", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("norm_eps", ctypes.c_float),
        ("global_rope_theta", ctypes.c_float),
        ("local_rope_theta", ctypes.c_float),
        ("chunk_rows", ctypes.c_int32),
        ("layer_is_global", ctypes.c_uint8 * OP_MAX_LAYERS),
        ("terms", ctypes.c_uint8 * 8),
        ("flags", ctypes.c_uint32),
        ("prune_pre_final_norm", ctypes.c_int32),
    ]


class OpCalibration(ctypes.Structure):
    _fields_ = [
        ("struct_bytes", ctypes.c_uint32),
        ("tolerance", ctypes.c_float),
        ("reference_set", ctypes.c_int32),
        ("default_set", ctypes.c_int32),
        ("chosen_set", ctypes.c_int32),
        ("n_candidates", ctypes.c_int32),
        ("candidate_set", ctypes.c_int32 * 16),
        ("candidate_err", ctypes.c_float * 16),
        ("n_rows", ctypes.c_int32),
        ("n_tokens", ctypes.c_int32),
        ("default_err", ctypes.c_float),
        ("flags", ctypes.c_uint32),
        ("mlp_layers_err", ctypes.c_float),
        ("mlp_layers", ctypes.c_uint64),
    ]


OP_CAL_FULL_REPORT = 1
OP_CAL_WHOLE_DEPTH = 2


class OpProfileEntry(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("launches", ctypes.c_int32), ("total_ms", ctypes.c_double)]


def library_path() -> Path:
    override = os.environ.get("OPEN_PROVENCE_HIP_LIB")
    if override:
        return Path(override)
    return Path(__file__).resolve().parent / LIB_NAME


_LIB: ctypes.CDLL | None = None


def load_library() -> ctypes.CDLL:
    """Load (once) and type the shared library; raises :class:`HipLibraryError` when it is absent."""

    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise HipLibraryError(
            f"{path} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback for the forward path."
        )
    try:
        lib = ctypes.CDLL(str(path))
    except OSError as exc:
        raise HipLibraryError(f"failed to load {path}: {exc}") from exc

    vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.op_abi_version.restype = ci
    lib.op_abi_version.argtypes = []
    lib.op_create.restype = ci
    lib.op_create.argtypes = [ctypes.POINTER(OpConfig), ctypes.POINTER(vp)]
    lib.op_load_weight.restype = ci
    lib.op_load_weight.argtypes = [vp, ctypes.c_char_p, vp, ci, ctypes.POINTER(ctypes.c_int64), ci]
    lib.op_weights_ready.restype = ci
    lib.op_weights_ready.argtypes = [vp]
    lib.op_workspace_bytes.restype = cs
    lib.op_workspace_bytes.argtypes = [vp, ci, ci, ci]
    lib.op_forward_packed.restype = ci
    lib.op_forward_packed.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, cs, vp]
    lib.op_effective_policy.restype = ci
    lib.op_effective_policy.argtypes = [vp, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ci)]
    if hasattr(lib, "op_set_compact_operands"):  # (absent only in an older library loaded for a same-box A/B, see below)
        lib.op_set_compact_operands.restype = ci
        lib.op_set_compact_operands.argtypes = [vp, ci, ctypes.POINTER(ci)]
    if hasattr(lib, "op_select_kernel_set"):
        lib.op_select_kernel_set.restype = ci
        lib.op_select_kernel_set.argtypes = [vp, ci]
        if hasattr(lib, "op_mlp_correction_layers"):
            lib.op_mlp_correction_layers.restype = ci
            lib.op_mlp_correction_layers.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
            lib.op_select_mlp_correction_layers.restype = ci
            lib.op_select_mlp_correction_layers.argtypes = [vp, ctypes.c_uint64]
        lib.op_calibrate.restype = ci
        lib.op_calibrate.argtypes = [vp, ctypes.c_float, vp, vp, ci, ctypes.POINTER(OpCalibration)]
    lib.op_segment_means.restype = ci
    lib.op_segment_means.argtypes = [vp, vp, ci, vp, ci, vp, vp]
    lib.op_debug_capture_hidden.restype = ci
    lib.op_debug_capture_hidden.argtypes = [vp, vp]
    lib.op_profile_enable.restype = ci
    lib.op_profile_enable.argtypes = [vp, ci]
    lib.op_profile_read.restype = ci
    lib.op_profile_read.argtypes = [vp, ctypes.POINTER(OpProfileEntry), ci]
    lib.op_profile_reset.restype = ci
    lib.op_profile_reset.argtypes = [vp]
    lib.op_profile_kind_name.restype = ctypes.c_char_p
    lib.op_profile_kind_name.argtypes = [ci]
    lib.op_debug_clock_probe.restype = ci
    lib.op_debug_clock_probe.argtypes = [vp, ci, vp, vp]
    lib.op_device_count.restype = ci
    lib.op_device_count.argtypes = [ctypes.POINTER(ci)]
    lib.op_destroy.restype = None
    lib.op_destroy.argtypes = [vp]
    lib.op_last_error.restype = ctypes.c_char_p
    lib.op_last_error.argtypes = [vp]

    version = lib.op_abi_version()
    # measurement hook: scripts/ab_lib.sh times an OLDER build of the library against the tree's (same box, alternating
    # runs); OPEN_PROVENCE_HIP_LIB_ANY_ABI=1 lets the explicitly named library through with its own version
    any_abi = os.environ.get("OPEN_PROVENCE_HIP_LIB") and os.environ.get("OPEN_PROVENCE_HIP_LIB_ANY_ABI") == "1"
    if version != OP_ABI_VERSION and not any_abi:
        raise HipLibraryError(f"{path} has ABI version {version}, the Python layer expects {OP_ABI_VERSION}; rebuild")
    _LIB = lib
    return lib


def last_error(lib: ctypes.CDLL, handle) -> str:
    raw = lib.op_last_error(handle)
    return raw.decode("utf-8", "replace") if raw else ""


def check(lib: ctypes.CDLL, handle, code: int, what: str) -> None:
    if code != OP_OK:
        raise HipLibraryError(f"{what} failed with code {code}: {last_error(lib, handle)}")
