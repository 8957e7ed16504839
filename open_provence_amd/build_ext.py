"""In-tree build of ``libopenprovence_hip.so`` with hipcc for gfx950 (cross-compiles without a GPU)."""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
SOURCES = [CSRC / "op_api.hip"]
HEADERS = [CSRC / "op_kernels.hip.h", PKG_DIR.parent / "include" / "open_provence_hip.h"]
OUTPUT = PKG_DIR / "libopenprovence_hip.so"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def is_stale() -> bool:
    if not OUTPUT.exists():
        return True
    built = OUTPUT.stat().st_mtime
    return any(src.stat().st_mtime > built for src in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source into the shared library next to the package; returns its path."""

    if not force and not is_stale():
        return OUTPUT
    cmd = [
        _hipcc(),
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-shared",
        "-Wno-unused-result",
        # no SLP vectorizer: it fuses adjacent scalar fp32 ops of the epilogues into v_pk_* instructions, which cost
        # more issue time beside MFMAs than the two scalar ops they replace (measured: +1.1 % pairs/s without it)
        "-fno-slp-vectorize",
        "-o",
        str(OUTPUT),
        *[str(s) for s in SOURCES],
    ]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    return OUTPUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
