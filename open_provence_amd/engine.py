"""``HipEncoder``: one handle of ``libopenprovence_hip.so`` bound to one GPU.

PyTorch-ROCm is used here for exactly three things -- owning device buffers (ids, outputs, workspace),
naming the current HIP stream, and reading checkpoint tensors -- never for arithmetic on the forward
path.  Everything numerical happens inside ``op_forward_packed`` (``include/open_provence_hip.h``).
"""

from __future__ import annotations

import ctypes
import os
from contextlib import contextmanager
from typing import Iterator, Mapping, Sequence

import numpy as np
import torch

from . import _lib
from .config import EncoderDims
from .packing import pack_rows

_PRECISIONS = {"bf16x3": _lib.OP_PRECISION_BF16X3, "bf16x2": _lib.OP_PRECISION_BF16X2, "bf16": _lib.OP_PRECISION_BF16}
# measurement / test switches, read ONCE when an encoder is created (the C ABI keeps no process-global state)
_ENV_FLAGS = {
    "OPEN_PROVENCE_FORCE_TILED": _lib.OP_FLAG_FORCE_TILED,
    "OPEN_PROVENCE_NO_SMALL_BLOCKS": _lib.OP_FLAG_NO_SMALL_BLOCKS,
    "OPEN_PROVENCE_NO_POLICY_KERNELS": _lib.OP_FLAG_NO_POLICY_KERNELS,
    "OPEN_PROVENCE_NO_LAYER_FUSION": _lib.OP_FLAG_NO_LAYER_FUSION,
    "OPEN_PROVENCE_LAYER_8X16": _lib.OP_FLAG_LAYER_8X16,
    "OPEN_PROVENCE_LAYER_M32": _lib.OP_FLAG_LAYER_M32,
    "OPEN_PROVENCE_NO_HEAD_FUSION": _lib.OP_FLAG_NO_HEAD_FUSION,
    "OPEN_PROVENCE_NO_F8": _lib.OP_FLAG_NO_F8,
    "OPEN_PROVENCE_ATTN_XCD_GROUP": _lib.OP_FLAG_ATTN_XCD_GROUP,
    "OPEN_PROVENCE_PANEL_F8": _lib.OP_FLAG_PANEL_F8,
    "OPEN_PROVENCE_PANEL_F8_WI": _lib.OP_FLAG_PANEL_F8_WI,
    "OPEN_PROVENCE_NO_LAYER_PAIRS": _lib.OP_FLAG_NO_LAYER_PAIRS,
}


def parse_precision(precision: "str | Mapping[str, int]") -> tuple[int, list[int]]:
    """``"bf16x3" | "bf16x2" | "bf16"`` or a per-family term-mask mapping / ``"wqkv=1,qk=3,..."`` string
    (families: ``_lib.OP_FAMILIES``; mask bit 0 = lo(activation) x hi, bit 1 = hi x lo(weight / key / value);
    families left out keep all terms) -> (enum op_precision, terms[8])."""

    terms = [3] * len(_lib.OP_FAMILIES) + [0] * (8 - len(_lib.OP_FAMILIES))
    if isinstance(precision, str) and precision in _PRECISIONS:
        return _PRECISIONS[precision], terms
    if isinstance(precision, str):
        try:
            mapping = {k.strip(): int(v) for k, v in (item.split("=") for item in precision.split(",") if item.strip())}
        except ValueError as exc:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)} or 'family=mask,...': {precision!r}") from exc
    else:
        mapping = dict(precision)
    for name, mask in mapping.items():
        if name not in _lib.OP_FAMILIES:
            raise ValueError(f"unknown contraction family {name!r}; expected one of {_lib.OP_FAMILIES}")
        if int(mask) not in (0, 1, 2, 3):
            raise ValueError(f"term mask of {name!r} must be 0..3, got {mask!r}")
        terms[_lib.OP_FAMILIES.index(name)] = int(mask)
    return _lib.OP_PRECISION_CUSTOM, terms
_DTYPES = {torch.float32: _lib.OP_DTYPE_F32, torch.bfloat16: _lib.OP_DTYPE_BF16, torch.float16: _lib.OP_DTYPE_F16}

DEFAULT_CALIBRATION_TOLERANCE = 1e-4  # max |logit difference| to the (hi, lo) bf16 kernels; the path's bar is 1e-3
PATH_TOLERANCE = 1e-3  # BASELINE.json north_star: logits within 1e-3 of the fp32 CPU reference


def resolve_calibration_tolerance(calibrate: "bool | float | None") -> float:
    """``False`` / ``0`` -> 0.0 (no calibration); a float -> that tolerance; ``True`` / ``None`` -> ``OPEN_PROVENCE_CALIBRATE``
    (``0`` / ``off`` disables, a number is the tolerance) or :data:`DEFAULT_CALIBRATION_TOLERANCE`."""

    if isinstance(calibrate, (bool, np.bool_)):  # (by TYPE: `1` / `np.True_` are not `True` by identity)
        if not calibrate:
            return 0.0
        calibrate = None
    if calibrate is not None:
        tolerance = float(calibrate)
        if isinstance(calibrate, (int, np.integer)) and tolerance == 1.0:
            calibrate = None  # `calibrate=1` means "on", not a tolerance of 1.0 logit
        elif not tolerance < PATH_TOLERANCE:  # (also NaN)
            raise ValueError(f"calibration tolerance {calibrate!r} is not inside the path's own bar ({PATH_TOLERANCE:g} on a logit): "
                             "any candidate would pass; use kernel_set= to pin a set regardless of its error")
        else:
            return max(tolerance, 0.0)
    env = os.environ.get("OPEN_PROVENCE_CALIBRATE", "").strip().lower()
    if env in ("0", "off", "false", "no"):
        return 0.0
    if env and env not in ("1", "on", "true", "yes"):
        try:
            tolerance = max(float(env), 0.0)
            if not tolerance < PATH_TOLERANCE:
                raise ValueError
            return tolerance
        except ValueError as exc:
            raise ValueError(f"OPEN_PROVENCE_CALIBRATE must be 0 / off or a tolerance below {PATH_TOLERANCE:g}, got {env!r}") from exc
    return DEFAULT_CALIBRATION_TOLERANCE


def require_gpu(device: torch.device | str | int | None = None) -> torch.device:
    """Resolve a HIP device or fail loudly (the product has no CPU path)."""

    if not torch.cuda.is_available():
        raise _lib.HipLibraryError(
            "No HIP device visible to PyTorch-ROCm: the OpenProvence MI355X path runs only on a GPU "
            "(there is no CPU fallback; the CPU restatement under oracle/ is test infrastructure)."
        )
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    if dev.type != "cuda":
        raise _lib.HipLibraryError(f"device {dev} is not a HIP device")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class HipEncoder:
    """ModernBERT cross-encoder + pruning/ranking heads as hand-written gfx950 kernels."""

    def __init__(
        self,
        dims: EncoderDims,
        *,
        device: torch.device | str | int | None = None,
        precision: "str | Mapping[str, int]" = "bf16x3",
        chunk_rows: int | None = None,
        prune_pre_final_norm: bool = False,
        flags: int | None = None,
    ) -> None:
        precision_code, terms = parse_precision(precision)
        self.lib = _lib.load_library()
        self.device = require_gpu(device)
        self.dims = dims
        self.precision = precision
        if dims.num_layers > _lib.OP_MAX_LAYERS:
            raise ValueError("too many layers")
        cfg = _lib.OpConfig()
        cfg.struct_bytes = ctypes.sizeof(_lib.OpConfig)
        cfg.device_id = int(self.device.index)
        cfg.vocab_size = dims.vocab_size
        cfg.hidden_size = dims.hidden_size
        cfg.intermediate_size = dims.intermediate_size
        cfg.num_layers = dims.num_layers
        cfg.num_heads = dims.num_heads
        cfg.num_labels = dims.num_labels
        cfg.local_attention = dims.local_attention
        cfg.max_position_embeddings = dims.max_position_embeddings
        cfg.pooling = _lib.OP_POOL_MEAN if dims.classifier_pooling == "mean" else _lib.OP_POOL_CLS
        cfg.precision = precision_code
        for i, mask in enumerate(terms):
            cfg.terms[i] = mask
        if flags is None:
            flags = 0
            for name, bit in _ENV_FLAGS.items():
                if os.environ.get(name):
                    flags |= bit
            waves = os.environ.get("OPEN_PROVENCE_ATT_WAVES")
            if waves:
                flags |= _lib.OP_FLAG_ATT_WAVES_4 if int(waves) == 4 else _lib.OP_FLAG_ATT_WAVES_8
        cfg.flags = int(flags)
        cfg.prune_pre_final_norm = 1 if prune_pre_final_norm else 0
        self.prune_pre_final_norm = bool(prune_pre_final_norm)
        cfg.norm_eps = dims.norm_eps
        cfg.global_rope_theta = dims.global_rope_theta
        cfg.local_rope_theta = dims.local_rope_theta
        cfg.chunk_rows = int(chunk_rows or 0)
        for i, flag in enumerate(dims.layer_is_global):
            cfg.layer_is_global[i] = 1 if flag else 0
        handle = ctypes.c_void_p()
        code = self.lib.op_create(ctypes.byref(cfg), ctypes.byref(handle))
        _lib.check(self.lib, None, code, "op_create")
        self._handle = handle
        self._split_state: dict | None = None  # the two pipeline streams + their workspaces (forward_packed_on)
        self._workspace: torch.Tensor | None = None
        self._capture: torch.Tensor | None = None
        self._capture_result: torch.Tensor | None = None
        self.calibration: dict | None = None  # report of the last calibrate() (load_state_dict runs it by default)

    # -- lifecycle -------------------------------------------------------------------------------
    def close(self) -> None:
        handle = getattr(self, "_handle", None)
        if handle is not None and handle.value:
            self.lib.op_destroy(handle)
            self._handle = ctypes.c_void_p()
        # (the two CU-masked streams of forward_packed_on stay alive for the life of the process: torch's caching allocator
        # keeps events on them for the outputs handed over with record_stream, and destroying a stream under it faults)
        self._split_state = None

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    # -- weights ---------------------------------------------------------------------------------
    def load_weight(self, name: str, tensor: torch.Tensor) -> None:
        t = tensor.detach()
        if t.dtype not in _DTYPES:
            t = t.to(torch.float32)
        t = t.contiguous()
        shape = (ctypes.c_int64 * max(t.ndim, 1))(*([int(s) for s in t.shape] or [1]))
        code = self.lib.op_load_weight(
            self._handle, name.encode("utf-8"), ctypes.c_void_p(t.data_ptr()), _DTYPES[t.dtype], shape, max(t.ndim, 1)
        )
        _lib.check(self.lib, self._handle, code, f"op_load_weight({name})")

    def load_state_dict(self, state: Mapping[str, torch.Tensor], *, calibrate: "bool | float | None" = None,
                        kernel_set: "str | None" = None, calibration_rows: "Sequence[Sequence[int]] | None" = None) -> None:
        """Checkpoint keys as in the reference's ``model.safetensors`` (``ranking_model.*`` /
        ``pruning_head.*``; legacy checkpoints without the prefix are accepted, standalone.py:1452-1464).
        Non-persistent buffers (``inv_freq``) and training-only tensors are skipped.

        Then the arithmetic is chosen FROM THE LOADED WEIGHTS (the reference decides its dtype and attention
        implementation at load time too, standalone.py:219-244, 1589-1615, 1631-1642): ``kernel_set`` (or
        ``OPEN_PROVENCE_KERNEL_SET``) pins a kernel set by name; otherwise :meth:`calibrate` runs with tolerance
        ``calibrate`` (a float; ``True`` / ``None`` = ``OPEN_PROVENCE_CALIBRATE`` or 1e-4; ``False`` / ``0`` = keep the
        default selection, which is safe for any weights and priced for the worst case)."""

        for name, tensor in state.items():
            if "inv_freq" in name or name.endswith("pooling_weights.weight") or name.endswith("pooling_weights.bias"):
                continue
            self.load_weight(name, tensor)
        _lib.check(self.lib, self._handle, self.lib.op_weights_ready(self._handle), "op_weights_ready")
        # A kernel set pinned or calibrated on the PREVIOUS checkpoint is not a property of this one: op_load_weight drops it
        # for every GEMM weight that changes (ABI 8); a state dict that reloads only norms / heads / embeddings keeps the
        # weights the set was measured on, but the measurement covered those tensors too -- back to the default selection.
        _lib.check(self.lib, self._handle, self.lib.op_select_kernel_set(self._handle, int(_lib.OP_KS_AUTO)), "op_select_kernel_set(auto)")
        self.__dict__["_f8_active"] = None
        self.__dict__["_audit_pending"] = False
        self.calibration = None
        pinned = kernel_set or os.environ.get("OPEN_PROVENCE_KERNEL_SET")
        if pinned:
            self.select_kernel_set(pinned)
            return
        tolerance = resolve_calibration_tolerance(calibrate)
        if tolerance > 0.0 and hasattr(self.lib, "op_calibrate"):
            self.calibrate(tolerance, rows=calibration_rows)

    def select_kernel_set(self, name: "str | None") -> None:
        """Pin the kernel set by its :meth:`effective_policy` name (``None`` / ``"auto"``: the default selection).  A set with
        fewer product terms than the checkpoint carries is an approximation: :meth:`calibrate` is what measures one."""

        number = _lib.OP_KS_AUTO if name in (None, "auto") else _lib.KERNEL_SET_IDS.get(str(name))
        if number is None:
            raise ValueError(f"unknown kernel set {name!r}; expected one of {sorted(_lib.KERNEL_SET_IDS)} or 'auto'")
        _lib.check(self.lib, self._handle, self.lib.op_select_kernel_set(self._handle, int(number)), f"op_select_kernel_set({name})")
        self.__dict__["_f8_active"] = None
        self.__dict__["_audit_pending"] = False  # (a pinned set is the caller's decision: nothing to audit)

    def _repin_calibrated(self, chosen: str) -> None:
        """Pin the calibrated set again after a detour through another one (the audits run the reference set in between):
        ``op_select_kernel_set`` means the whole depth, so the layer mask of sets 8 / 9 is pinned again behind it."""

        self.select_kernel_set(chosen)
        layers = (self.calibration or {}).get("mlp_correction_layers")
        if layers is not None and chosen in ("f16+mlp-f16-f8-w", "f16+mlp-f16-f8") and chosen == (self.calibration or {}).get("chosen_set"):
            mask = sum(1 << int(li) for li in layers)
            _lib.check(self.lib, self._handle, self.lib.op_select_mlp_correction_layers(self._handle, ctypes.c_uint64(mask)),
                       "op_select_mlp_correction_layers")

    def calibrate(self, tolerance: float = 1e-4, rows: "Sequence[Sequence[int]] | None" = None, *, full_report: "bool | None" = None,
                  whole_depth: "bool | None" = None) -> dict:
        """``op_calibrate``: one batch (``rows`` of token ids -- a sample of real inputs -- or the library's synthetic
        batch) through the (hi, lo) bf16 kernels and through every kernel set cheaper than the default one; the
        cheapest whose logits stay within ``tolerance`` of them (and finite) is what the forward runs on from now on.
        Returns (and keeps as ``self.calibration``) the report: ``{"tolerance", "reference_set", "default_set",
        "chosen_set", "candidates": {set name: max |logit difference|}, "rows", "tokens", "batch"}`` -- plus, when the choice is
        kernel set 8 / 9, ``"mlp_correction_layers"`` (the layers that keep the fp16 + e4m3 MLP) and ``"mlp_correction_err"``."""

        report = _lib.OpCalibration()
        report.struct_bytes = ctypes.sizeof(_lib.OpCalibration)
        # cheapest first, stop at the first candidate that holds (up to 11 fewer forwards at load on the deep models);
        # full_report=True / OPEN_PROVENCE_CALIBRATE_FULL=1 measures every candidate (scripts/calibration_probe.py)
        if full_report is None:
            full_report = os.environ.get("OPEN_PROVENCE_CALIBRATE_FULL", "").strip().lower() in ("1", "on", "true", "yes")
        report.flags = _lib.OP_CAL_FULL_REPORT if full_report else 0
        # kernel sets 8 / 9 are refined layer by layer (ABI 9); whole_depth=True / OPEN_PROVENCE_CALIBRATE_WHOLE_DEPTH=1 keeps them whole
        if whole_depth is None:
            whole_depth = os.environ.get("OPEN_PROVENCE_CALIBRATE_WHOLE_DEPTH", "").strip().lower() in ("1", "on", "true", "yes")
        if whole_depth:
            report.flags |= _lib.OP_CAL_WHOLE_DEPTH
        if rows is not None:
            ids_np, cu_np, _ = pack_rows(rows)
            self.check_ids(ids_np)
            ids_np = np.ascontiguousarray(ids_np, dtype=np.int32)
            cu_np = np.ascontiguousarray(cu_np, dtype=np.int32)
            args = (ids_np.ctypes.data_as(ctypes.c_void_p), cu_np.ctypes.data_as(ctypes.c_void_p), int(cu_np.shape[0]) - 1)
        else:
            args = (None, None, 0)
        with torch.cuda.device(self.device):
            code = self.lib.op_calibrate(self._handle, ctypes.c_float(float(tolerance)), *args, ctypes.byref(report))
        _lib.check(self.lib, self._handle, code, "op_calibrate")
        names = _lib.KERNEL_SET_NAMES
        self.__dict__["_f8_active"] = None
        self.calibration = {
            "tolerance": float(report.tolerance),
            "reference_set": names.get(int(report.reference_set), str(report.reference_set)),
            "default_set": names.get(int(report.default_set), str(report.default_set)),
            "chosen_set": names.get(int(report.chosen_set), str(report.chosen_set)),
            "candidates": {names.get(int(report.candidate_set[i]), str(report.candidate_set[i])): float(report.candidate_err[i])
                           for i in range(int(report.n_candidates))},
            "default_err": float(report.default_err),
            "rows": int(report.n_rows),
            "tokens": int(report.n_tokens),
            "batch": "caller rows" if rows is not None else "synthetic (uniform token ids)",
        }
        # kernel sets 8 / 9 layer by layer (ABI 9): the layers that keep the fp16 + e4m3 MLP, the others run the "f16" set's
        if self.calibration["chosen_set"] in ("f16+mlp-f16-f8-w", "f16+mlp-f16-f8"):
            mask = int(report.mlp_layers)
            self.calibration["mlp_correction_layers"] = [li for li in range(self.dims.num_layers) if (mask >> li) & 1]
            self.calibration["mlp_correction_err"] = float(report.mlp_layers_err)
        # a set chosen on SYNTHETIC token ids is audited on the first real batch (_audit_first_batch); the caller's own rows
        # are real inputs already.  OPEN_PROVENCE_AUDIT=0 switches the audit off.
        self.__dict__["_audit_pending"] = (rows is None and self.calibration["chosen_set"] != self.calibration["default_set"]
                                           and os.environ.get("OPEN_PROVENCE_AUDIT", "1").strip().lower() not in ("0", "off", "false", "no"))
        return self.calibration

    def effective_policy(self) -> dict:
        """Term masks actually evaluated (the requested policy minus weight-lo terms that are identically zero
        for the loaded checkpoint) and the kernel set running them: ``{"terms": {family: mask}, "kernel_set":
        "bf16x3" | "bf16-weights" | "bf16" | "f16-f8" | "f16-f8-w" | "all-terms kernels, cleared lo operands"}`` ("f16-f8":
        the terms of "bf16-weights" with the whole-layer kernel's operands carried as fp16 hi + e4m3 lo; "f16-f8-w": the
        terms of "bf16x3" in that format, the weights' lo part as a third plane)."""

        terms = (ctypes.c_uint8 * 8)()
        kernel_set = ctypes.c_int(0)
        code = self.lib.op_effective_policy(self._handle, terms, ctypes.byref(kernel_set))
        _lib.check(self.lib, self._handle, code, "op_effective_policy")
        names = _lib.KERNEL_SET_NAMES
        policy = {
            "terms": {name: int(terms[i]) for i, name in enumerate(_lib.OP_FAMILIES)},
            "kernel_set": names.get(int(kernel_set.value), str(kernel_set.value)),
        }
        if policy["kernel_set"] in ("f16+mlp-f16-f8-w", "f16+mlp-f16-f8") and hasattr(self.lib, "op_mlp_correction_layers"):
            mask = ctypes.c_uint64(0)
            code = self.lib.op_mlp_correction_layers(self._handle, ctypes.byref(mask))
            _lib.check(self.lib, self._handle, code, "op_mlp_correction_layers")
            policy["mlp_correction_layers"] = [li for li in range(self.dims.num_layers) if (int(mask.value) >> li) & 1]
        return policy

    # -- the range guard of the fp16 + e4m3 kernel sets ---------------------------------------------
    def f8_active(self) -> bool:
        """True while the forward runs on kernel sets 3 / 4 (fp16 hi + e4m3 lo operands): their fp16 plane has fp16's
        range, and an MLP activation beyond it comes out as NaN by design (never clamped)."""

        cached = self.__dict__.get("_f8_active")
        if cached is None:
            cached = self.__dict__["_f8_active"] = self.effective_policy()["kernel_set"] in _lib.FP16_PLANE_SETS
        return cached

    def fall_back_from_f8(self, reason: str = "") -> bool:
        """Switch this model to the (hi, lo) bf16 kernel sets for good (``op_set_compact_operands``: both weight packs are resident,
        nothing is re-loaded) and warn once.  Returns True when the kernel set changed, i.e. when repeating the forward
        can give a different answer.  The reference's own precedent for a silent, correct retry: its fallback from an
        unsupported dtype / attention implementation at load time (standalone.py:1631-1642)."""

        if not self.f8_active() or not hasattr(self.lib, "op_set_compact_operands"):
            return False
        changed = ctypes.c_int(0)
        code = self.lib.op_set_compact_operands(self._handle, 0, ctypes.byref(changed))
        _lib.check(self.lib, self._handle, code, "op_set_compact_operands")
        self.__dict__["_f8_active"] = None
        if changed.value:
            import warnings

            if self.calibration is not None:  # the report names what RUNS, not what was chosen before the fallback
                self.calibration["chosen_set"] = self.effective_policy()["kernel_set"]
                self.calibration["fallback"] = "fp16 range guard"
            self.__dict__["_audit_pending"] = False

            self.fallbacks = int(getattr(self, "fallbacks", 0)) + 1  # reported by process() (timing / performance_trace)

            warnings.warn(
                "open_provence_amd: a forward on the fp16 + e4m3 kernel set returned non-finite values"
                + (f" ({reason})" if reason else "")
                + ": an activation left fp16's range.  The batch is repeated on the (hi, lo) bf16 kernels (fp32 range) and "
                "this model stays on them (slower: 2-3 instead of 1.5-2 MFMA products per term); set OPEN_PROVENCE_NO_F8=1 "
                "to start there.",
                RuntimeWarning, stacklevel=3,
            )
        return bool(changed.value)

    def forward_packed_checked(self, ids, cu_seqlens, cu_seqlens_host, max_seqlen, keep_prob=None):
        """``forward_packed`` + the range guard: on kernel sets 3 / 4 the outputs are tested for NaN / Inf (one device
        reduction, one synchronisation) and a non-finite batch is repeated on the (hi, lo) bf16 sets.  On those sets
        nothing is tested: whatever comes out is what the reference's arithmetic gives."""

        prune, rank = self.forward_packed(ids, cu_seqlens, cu_seqlens_host, max_seqlen, keep_prob=keep_prob)
        if self.f8_active():
            ok = torch.isfinite(rank).all() & torch.isfinite(prune).all()
            if not bool(ok.item()) and self.fall_back_from_f8("forward"):
                prune, rank = self.forward_packed(ids, cu_seqlens, cu_seqlens_host, max_seqlen, keep_prob=keep_prob)
        return prune, rank

    # -- forward ---------------------------------------------------------------------------------
    def _ensure_workspace(self, n_seqs: int, total_tokens: int, max_seqlen: int) -> torch.Tensor:
        need = int(self.lib.op_workspace_bytes(self._handle, n_seqs, total_tokens, max_seqlen))
        if need <= 0:
            raise _lib.HipLibraryError("op_workspace_bytes returned 0")
        ws = self._workspace
        if ws is None or ws.numel() < need:
            self._workspace = None
            ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._workspace = ws
        return ws

    def segment_means(self, values: torch.Tensor, segments: torch.Tensor) -> torch.Tensor:
        """``values[T]`` fp32 and ``segments[S, 2]`` int32 (token ranges ``[start, end)``) on this device ->
        ``[S]`` fp32: ``values[start:end].mean()`` in numpy's float32 pairwise order, bit for bit; 1.0 for an empty
        range (the reference's per-fragment score, standalone.py:3075-3082).  Asynchronous on the current stream."""

        if values.dtype != torch.float32 or segments.dtype != torch.int32:
            raise TypeError("values must be fp32 and segments int32")
        if values.device != self.device or segments.device != self.device:
            raise ValueError(f"values / segments must live on {self.device}")
        if not values.is_contiguous() or not segments.is_contiguous() or segments.ndim != 2 or segments.shape[1] != 2:
            raise ValueError("values must be contiguous and segments a contiguous [S, 2] tensor")
        n_seg = int(segments.shape[0])
        out = torch.empty(n_seg, dtype=torch.float32, device=self.device)
        if n_seg == 0:
            return out
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            code = self.lib.op_segment_means(
                self._handle, ctypes.c_void_p(values.data_ptr()), int(values.numel()), ctypes.c_void_p(segments.data_ptr()),
                n_seg, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream),
            )
        _lib.check(self.lib, self._handle, code, "op_segment_means")
        return out

    def forward_packed(
        self,
        ids: torch.Tensor,
        cu_seqlens: torch.Tensor,
        cu_seqlens_host: np.ndarray,
        max_seqlen: int,
        keep_prob: torch.Tensor | None = None,
    ) -> tuple[torch.Tensor, torch.Tensor]:
        """``ids[T]`` / ``cu_seqlens[B+1]`` int32 on this device -> (prune_logits[T, 2], rank_logits[B, nl]) fp32.
        ``keep_prob`` (optional, fp32 ``[T]`` on this device) additionally receives
        ``softmax(prune_logits, -1)[:, 1]``, evaluated in the head kernel.

        Asynchronous on the current torch stream of ``self.device``."""

        total, n_seqs = self._check_packed_inputs(ids, cu_seqlens, keep_prob)
        cu_host = np.ascontiguousarray(cu_seqlens_host, dtype=np.int32)
        if cu_host.shape[0] != n_seqs + 1:
            raise ValueError("cu_seqlens_host length mismatch")
        prune = torch.empty((total, 2), dtype=torch.float32, device=self.device)
        rank = torch.empty((n_seqs, self.dims.num_labels), dtype=torch.float32, device=self.device)
        if n_seqs == 0:
            return prune, rank
        ws = self._ensure_workspace(n_seqs, total, int(max_seqlen))
        if self._capture is not None:
            self._capture = torch.zeros(
                (self.dims.num_layers + 1, total, self.dims.hidden_size), dtype=torch.float32, device=self.device
            )
            capture_ptr = ctypes.c_void_p(self._capture.data_ptr())
            _lib.check(self.lib, self._handle, self.lib.op_debug_capture_hidden(self._handle, capture_ptr), "capture")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._forward_native(ids.data_ptr(), cu_seqlens.data_ptr(), cu_host, n_seqs, total, int(max_seqlen),
                                 prune.data_ptr(), rank.data_ptr(), keep_prob.data_ptr() if keep_prob is not None else None,
                                 ws, stream)
            self._maybe_audit(ids, cu_seqlens, cu_host, n_seqs, total, int(max_seqlen), prune, rank, keep_prob, ws, stream)
        return prune, rank

    # -- the audit under a process group: ONE verdict for all ranks (sharding.collective_audit) -----------------------
    @property
    def audit_pending(self) -> bool:
        """True while a kernel set chosen on the library's synthetic batch has not seen real rows yet."""

        return bool(self.__dict__.get("_audit_pending"))

    def audit_rows(self, rows: "Sequence[Sequence[int]]") -> "bool | None":
        """The first-real-batch audit as an explicit call that only MEASURES: ``rows`` through the calibrated set and through the
        reference set, ``calibration["audit"]`` filled, the pending flag cleared; the kernel set is left where it is (the
        caller -- ``sharding.collective_audit`` -- combines the ranks' verdicts and reverts all of them or none).  Returns the
        verdict, or None when there is nothing to audit (no calibrated set pending, or fewer than 64 tokens)."""

        if not self.audit_pending:
            return None
        ids_np, cu_np, max_len = pack_rows(rows)
        total, n_seqs = int(cu_np[-1]), len(cu_np) - 1
        if total < 64:
            return None
        self.check_ids(ids_np)
        ids = torch.from_numpy(ids_np).to(self.device)
        cu = torch.from_numpy(cu_np).to(self.device)
        cal = self.calibration or {}
        chosen, reference = cal.get("chosen_set"), cal.get("reference_set")
        self.__dict__["_audit_pending"] = False
        if not chosen or chosen == cal.get("default_set") or reference not in _lib.KERNEL_SET_IDS:
            return None
        outs = {}
        outs[chosen] = self.forward_packed(ids, cu, cu_np, max_len)  # (what is pinned now: the calibrated set, its layer mask included)
        _lib.check(self.lib, self._handle, self.lib.op_select_kernel_set(self._handle, int(_lib.KERNEL_SET_IDS[reference])), "op_select_kernel_set")
        outs[reference] = self.forward_packed(ids, cu, cu_np, max_len)
        self._repin_calibrated(chosen)
        self.__dict__["_f8_active"] = None
        err = float(torch.maximum((outs[chosen][0] - outs[reference][0]).abs().max(), (outs[chosen][1] - outs[reference][1]).abs().max()).item())
        bound = float(cal.get("tolerance", DEFAULT_CALIBRATION_TOLERANCE)) * float(getattr(self, "audit_factor", 3.0))
        passed = err == err and err <= bound
        cal["audit"] = {"tokens": total, "rows": n_seqs, "max_abs_err": err, "bound": bound, "passed": bool(passed), "collective": True}
        return bool(passed)

    def revert_to_default(self, reason: str) -> str:
        """Back to the default selection of ``op_weights_ready`` for good (what a failed audit does), with a warning."""

        import warnings

        before = self.effective_policy()["kernel_set"]
        self.select_kernel_set("auto")
        after = self.effective_policy()["kernel_set"]
        if self.calibration is not None:
            self.calibration["chosen_set"] = after
            self.calibration["reverted"] = reason
        if after != before:
            warnings.warn(f"open_provence_amd: kernel set {before!r} dropped ({reason}); this model runs on {after!r} from now on.",
                          RuntimeWarning, stacklevel=3)
        return after

    def _maybe_audit(self, ids, cu_seqlens, cu_host, n_seqs, total, max_seqlen, prune, rank, keep_prob, ws, stream) -> None:
        """The first-real-batch audit of a synthetically calibrated kernel set, from EITHER forward entry point
        (``forward_packed`` / ``forward_packed_on``).  Skipped -- and left pending -- for batches under 64 tokens, while hidden
        states are captured, and while the stream is being captured into a hipGraph (the audit synchronises and switches the
        handle's kernel set in the middle of the forward)."""

        if not self.__dict__.get("_audit_pending") or total < 64 or self._capture is not None:
            return
        if getattr(self, "audit_collective", False):  # under a process group the ranks audit TOGETHER (sharding.collective_audit)
            return
        if torch.cuda.is_current_stream_capturing():
            return
        self._audit_first_batch(ids, cu_seqlens, cu_host, n_seqs, total, max_seqlen, prune, rank, keep_prob, ws, stream)

    def _audit_first_batch(self, ids, cu_seqlens, cu_host, n_seqs, total, max_seqlen, prune, rank, keep_prob, ws, stream) -> None:
        """The calibration ran on synthetic token ids; the FIRST real batch a calibrated model sees is its audit: the same
        batch once more through the reference kernel set of the calibration, max |logit difference| against what the chosen
        set just returned.  Within ``audit_factor`` (3) x the calibration tolerance -- 3e-4, still 3 x inside the path's bar;
        the forward fuzz puts the worst row of other inputs at <= 2.7 x a calibration batch's maximum -- the choice stands
        (one synchronisation, two extra forwards, once per load).  Beyond it, or non-finite: the model goes back to the
        default selection of ``op_weights_ready`` for good, warns, and THIS batch is recomputed there before it is returned."""

        self.__dict__["_audit_pending"] = False
        cal = self.calibration or {}
        chosen, reference = cal.get("chosen_set"), cal.get("reference_set")
        if not chosen or chosen == cal.get("default_set") or reference not in _lib.KERNEL_SET_IDS:
            return
        p_ref, r_ref = torch.empty_like(prune), torch.empty_like(rank)
        profiling = bool(self.__dict__.get("_profiling"))
        if profiling:  # the audit's launches are not the caller's workload: keep them out of a per-kernel profile
            self.lib.op_profile_enable(self._handle, 0)
        self.select_kernel_set(reference)
        try:
            self._forward_native(ids.data_ptr(), cu_seqlens.data_ptr(), cu_host, n_seqs, total, max_seqlen,
                                 p_ref.data_ptr(), r_ref.data_ptr(), None, ws, stream)
        finally:
            self._repin_calibrated(chosen)
            if profiling:
                self.lib.op_profile_enable(self._handle, 1)
        err = float(torch.maximum((prune - p_ref).abs().max(), (rank - r_ref).abs().max()).item())  # (synchronises)
        bound = float(cal.get("tolerance", DEFAULT_CALIBRATION_TOLERANCE)) * float(getattr(self, "audit_factor", 3.0))
        passed = err == err and err <= bound  # (NaN fails)
        cal["audit"] = {"tokens": int(total), "rows": int(n_seqs), "max_abs_err": err, "bound": bound, "passed": bool(passed)}
        if passed:
            return
        import warnings

        self.select_kernel_set("auto")
        cal["chosen_set"] = self.effective_policy()["kernel_set"]
        warnings.warn(
            f"open_provence_amd: the first real batch disagrees with the load-time calibration: kernel set {chosen!r} is {err:.2e} from "
            f"the {reference!r} kernels on it (bound {bound:.1e}); this model runs on {cal['chosen_set']!r} from now on.  "
            "Pass calibration_rows= (a sample of real token ids) to calibrate on representative inputs.",
            RuntimeWarning, stacklevel=4,
        )
        self._forward_native(ids.data_ptr(), cu_seqlens.data_ptr(), cu_host, n_seqs, total, max_seqlen,
                             prune.data_ptr(), rank.data_ptr(), keep_prob.data_ptr() if keep_prob is not None else None, ws, stream)

    def _check_packed_inputs(self, ids: torch.Tensor, cu_seqlens: torch.Tensor, keep_prob: torch.Tensor | None) -> tuple[int, int]:
        """Shared argument checks of forward_packed / forward_packed_on -> (total_tokens, n_seqs)."""

        if ids.dtype != torch.int32 or cu_seqlens.dtype != torch.int32:
            raise TypeError("ids and cu_seqlens must be int32")
        if ids.device != self.device or cu_seqlens.device != self.device:
            raise ValueError(f"ids/cu_seqlens must live on {self.device}")
        total = int(ids.numel())
        if keep_prob is not None and (
            keep_prob.dtype != torch.float32 or keep_prob.device != self.device or keep_prob.numel() != total
            or not keep_prob.is_contiguous()
        ):
            raise ValueError("keep_prob must be a contiguous fp32 tensor of total_tokens elements on the encoder's device")
        return total, int(cu_seqlens.numel()) - 1

    def _forward_native(self, ids_ptr, cu_ptr, cu_host: np.ndarray, n_seqs: int, total: int, max_seqlen: int,
                        prune_ptr, rank_ptr, keep_ptr, ws: torch.Tensor, stream: int) -> None:
        base = ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        code = self.lib.op_forward_packed(
            self._handle,
            ctypes.c_void_p(ids_ptr),
            ctypes.c_void_p(cu_ptr),
            cu_host.ctypes.data_as(ctypes.c_void_p),
            n_seqs,
            total,
            max_seqlen,
            ctypes.c_void_p(prune_ptr),
            ctypes.c_void_p(rank_ptr),
            ctypes.c_void_p(keep_ptr) if keep_ptr is not None else None,
            ctypes.c_void_p(aligned),
            ctypes.c_size_t(ws.numel() - (aligned - base)),
            ctypes.c_void_p(stream),
        )
        _lib.check(self.lib, self._handle, code, "op_forward_packed")

    def _split_streams(self) -> dict:
        """Two HIP streams of their own (hipExtStreamCreateWithCUMask with the full mask: own hardware queues, no CU partition since
        round 6; plain streams if the runtime refuses) and a workspace per stream, created once."""

        st = self._split_state
        if st is None:
            n_cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
            streams = []
            raw_streams: list[int] = []
            hip = None
            try:
                hip = ctypes.CDLL("libamdhip64.so")
                words = (n_cus + 31) // 32
                with torch.cuda.device(self.device):
                    for half in range(2):
                        # Which CUs a pipeline gets (round 6): ALL of them -- two streams created through the CU-mask API (their
                        # own hardware queues) with the full mask.  Halves of the chip (contiguous halves of the mask's bit index =
                        # 16 CUs of every XCD each: rounds 2 - 5) leave the two launch sequences in one of two phase relations for
                        # a whole run, the bad one BELOW one sequence (63.6 - 64.3 k pairs/s in 4 of 5 hundred-step runs on one
                        # box, 68.8 k on another); unpartitioned, the blocks of the two sequences share the CUs as they come --
                        # 66.8 - 69.5 k on both boxes, never below (profiles/r06_exp_phase_diversity.txt, section 6; which CUs a
                        # mask really gives: microbench/cu_mask_probe.hip -- alternate bits or runs of 4 are NOT honoured as a
                        # partition, both streams get all 256 CUs).  OPEN_PROVENCE_PIPELINE_MASK_GROUP: 0 = the contiguous
                        # halves, g > 0 = runs of g bits alternate (measurement hook).
                        group = int(os.environ.get("OPEN_PROVENCE_PIPELINE_MASK_GROUP", "-1") or 0)
                        bits = [True if group < 0 else ((c // group) % 2 if group > 0 else (c * 2 // n_cus)) == half for c in range(n_cus)]
                        mask = (ctypes.c_uint32 * words)(
                            *[sum(1 << b for b in range(32) if w * 32 + b < n_cus and bits[w * 32 + b]) for w in range(words)]
                        )
                        handle = ctypes.c_void_p()
                        if hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask) != 0:
                            raise OSError("hipExtStreamCreateWithCUMask failed")
                        raw_streams.append(int(handle.value))
                        streams.append(torch.cuda.ExternalStream(handle.value, device=self.device))
            except (OSError, AttributeError):
                streams = [torch.cuda.Stream(self.device) for _ in range(2)]
            st = self._split_state = {"streams": streams, "ws": [None, None], "raw_streams": raw_streams[: len(streams)] if len(raw_streams) == len(streams) else [], "hip": hip}
        return st

    def forward_packed_on(
        self,
        part: int,
        ids: torch.Tensor,
        cu_seqlens: torch.Tensor,
        cu_seqlens_host: np.ndarray,
        max_seqlen: int,
        keep_prob: torch.Tensor | None = None,
    ) -> tuple[torch.Tensor, torch.Tensor]:
        """``forward_packed`` enqueued on pipeline ``part`` (0 or 1): its own HIP stream (own hardware queue; the whole chip
        since round 6, see ``_split_streams``) and its own workspace -- nothing is ordered against the caller's current stream or the other
        pipeline (inputs must already be resident; wait on ``pipeline_stream(part)`` before reading the outputs).

        Why two pipelines: every CU of a launch is in the same phase at the same time -- all fetch, then all multiply
        -- so HBM idles while the matrix pipes work and vice versa; two INDEPENDENT launch sequences
        drift apart and fill each other's gaps (xsmall, 2 x 128 pairs x 512 against 1 x 256: +3 .. +6 % pairs/s, same box).
        Forking and joining the halves inside every forward instead re-aligns them each time and loses 4 %."""

        if part not in (0, 1):
            raise ValueError("part must be 0 or 1")
        if self._capture is not None:
            raise RuntimeError("hidden-state capture is not available on the pipelined path (use forward_packed)")
        total, n_seqs = self._check_packed_inputs(ids, cu_seqlens, keep_prob)
        cu_host = np.ascontiguousarray(cu_seqlens_host, dtype=np.int32)
        if cu_host.shape[0] != n_seqs + 1:
            raise ValueError("cu_seqlens_host length mismatch")
        st = self._split_streams()
        side = st["streams"][part]
        # The outputs come from the SIDE stream's pool of the caching allocator (allocated under that stream): a block of
        # that pool is only ever recycled in side-stream order, so the kernels that write them can never land on memory a
        # still-queued reader of the caller's stream owns (allocated from the caller's pool, a block freed there and
        # whose last reader is still queued could be handed out here and overwritten early: nothing orders the side
        # stream behind the caller's).  Consumers wait on pipeline_stream(part) before reading; a consumer that reads
        # them on ANOTHER stream and drops them right away should record_stream() them there, as with any tensor that
        # crosses streams.
        with torch.cuda.device(self.device), torch.cuda.stream(side):
            prune = torch.empty((total, 2), dtype=torch.float32, device=self.device)
            rank = torch.empty((n_seqs, self.dims.num_labels), dtype=torch.float32, device=self.device)
            if n_seqs == 0:
                return prune, rank
            need = int(self.lib.op_workspace_bytes(self._handle, n_seqs, total, int(max_seqlen)))
            ws = st["ws"][part]
            if ws is None or ws.numel() < need + 256:
                ws = st["ws"][part] = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            self._forward_native(ids.data_ptr(), cu_seqlens.data_ptr(), cu_host, n_seqs, total, int(max_seqlen),
                                 prune.data_ptr(), rank.data_ptr(), keep_prob.data_ptr() if keep_prob is not None else None,
                                 ws, side.cuda_stream)
            # (the audit synchronises this pipeline's stream once; the other pipeline must not be mid-forward on another
            # host thread while it runs -- the pipelines of one encoder are driven from ONE thread everywhere in this package)
            self._maybe_audit(ids, cu_seqlens, cu_host, n_seqs, total, int(max_seqlen), prune, rank, keep_prob, ws, side.cuda_stream)
        return prune, rank

    def pipeline_stream(self, part: int) -> torch.cuda.Stream:
        return self._split_streams()["streams"][part]

    def check_ids(self, ids: np.ndarray) -> None:
        """``nn.Embedding`` raises on out-of-range ids (the reference's behaviour); the kernel only clamps them as a
        memory-safety net, so the host validates wherever ids are still in host memory."""

        if ids.size and (int(ids.min()) < 0 or int(ids.max()) >= self.dims.vocab_size):
            raise IndexError(
                f"token id out of range for the embedding table: ids span [{int(ids.min())}, {int(ids.max())}], "
                f"vocab_size is {self.dims.vocab_size}"
            )

    def forward_rows(self, rows: Sequence[Sequence[int]]) -> tuple[torch.Tensor, torch.Tensor, np.ndarray]:
        """Convenience: host id rows -> one H2D copy -> forward.  Returns (prune[T,2], rank[B,nl], cu_host)."""

        ids_np, cu_np, max_len = pack_rows(rows)
        self.check_ids(ids_np)
        ids = torch.from_numpy(ids_np).to(self.device, non_blocking=False)
        cu = torch.from_numpy(cu_np).to(self.device, non_blocking=False)
        prune, rank = self.forward_packed(ids, cu, cu_np, max_len)
        return prune, rank, cu_np

    # -- test / measurement hooks ------------------------------------------------------------------
    @contextmanager
    def capture_hidden(self) -> Iterator[None]:
        """While active, each forward also stores the N+1 hidden states; read :attr:`captured`."""

        self._capture = torch.empty(0)
        try:
            yield
        finally:
            self.lib.op_debug_capture_hidden(self._handle, None)
            self._capture_result = self._capture
            self._capture = None

    @property
    def captured(self) -> torch.Tensor:
        return self._capture_result

    def clock_probe(self, spin_us: int) -> "tuple[torch.Tensor, torch.cuda.Stream]":
        """Start a one-wave probe on a stream of its own that spins for ``spin_us`` microseconds beside whatever runs
        meanwhile; returns (uint64 tensor [shader cycles, 100 MHz ticks], its stream).  After synchronising that stream,
        ``cycles / ticks / 10`` = the shader clock in GHz the chip held (measurement hook: bench.py)."""

        with torch.cuda.device(self.device):
            side = torch.cuda.Stream(self.device)
            out = torch.zeros(2, dtype=torch.int64, device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            code = self.lib.op_debug_clock_probe(self._handle, int(spin_us), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(side.cuda_stream))
        _lib.check(self.lib, self._handle, code, "op_debug_clock_probe")
        return out, side

    def profile_enable(self, enabled: bool) -> None:
        _lib.check(self.lib, self._handle, self.lib.op_profile_enable(self._handle, 1 if enabled else 0), "profile")
        self.__dict__["_profiling"] = bool(enabled)

    def profile_reset(self) -> None:
        _lib.check(self.lib, self._handle, self.lib.op_profile_reset(self._handle), "profile_reset")

    def profile_read(self) -> dict[str, dict[str, float]]:
        """Kernel kind -> {launches, total_ms, avg_ms}; HIP events recorded on the launch stream."""

        entries = (_lib.OpProfileEntry * 32)()
        n = self.lib.op_profile_read(self._handle, entries, 32)
        if n < 0:
            _lib.check(self.lib, self._handle, n, "op_profile_read")
        out: dict[str, dict[str, float]] = {}
        for i in range(n):
            e = entries[i]
            name = self.lib.op_profile_kind_name(e.kind).decode()
            out[name] = {
                "launches": int(e.launches),
                "total_ms": float(e.total_ms),
                "avg_ms": float(e.total_ms) / max(int(e.launches), 1),
            }
        return out
