"""In-tree build of ``libopenprovence_hip.so`` with hipcc for gfx950 (cross-compiles without a GPU).

The library is several translation units -- the C ABI plus one unit per kernel family, each instantiating its
templates for every curated precision policy -- compiled in parallel into ``build/`` and linked into one shared
object next to the package.
"""

from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include" / "open_provence_hip.h"
BUILD_DIR = PKG_DIR.parent / "build" / "hip"
OUTPUT = PKG_DIR / "libopenprovence_hip.so"

_COMMON = [CSRC / "opk_common.hip.h"]
# opk_rowgemm.hip.h = parameters / pack kernels + stream helpers + the kernel, whose body is cut by phase into .inc files
_ROWGEMM = [CSRC / name for name in ("opk_rowgemm.hip.h", "opk_rowgemm_pack.hip.h", "opk_rowgemm_stream.hip.h", "opk_rowgemm_ln.hip.h", "opk_kstream.hip.h",
                                     "opk_rowgemm_phase1.hip.h", "opk_rowgemm_mlp.hip.h", "opk_rowgemm_mlp_ops.inc", "opk_rowgemm_mlp_loop.inc",
                                     "opk_rowgemm_qkv_pairs.hip.h", "opk_rowgemm_chunks.hip.h")]
_INTERNAL = _COMMON + [CSRC / "op_internal.h", CSRC / "opk_attn.hip.h", CSRC / "opk_panel.hip.h", CSRC / "opk_layer32.hip.h", CSRC / "opk_layer16p.hip.h"] + _ROWGEMM

# (object name, source, extra defines, headers it depends on)
UNITS = [
    ("op_api", CSRC / "op_api.hip", [], _INTERNAL + [CSRC / "opk_small.hip.h", CSRC / "opk_tiled.hip.h", INCLUDE]),
    ("op_launch_row0", CSRC / "op_launch_row.hip", ["-DOPL_ROW_PART=0"], _INTERNAL),
    ("op_launch_row1", CSRC / "op_launch_row.hip", ["-DOPL_ROW_PART=1"], _INTERNAL),
    ("op_launch_row2", CSRC / "op_launch_row.hip", ["-DOPL_ROW_PART=2"], _INTERNAL),
    ("op_launch_row3", CSRC / "op_launch_row.hip", ["-DOPL_ROW_PART=3"], _INTERNAL),
    ("op_launch_row4", CSRC / "op_launch_row.hip", ["-DOPL_ROW_PART=4"], _INTERNAL),
    ("op_launch_row5", CSRC / "op_launch_row.hip", ["-DOPL_ROW_PART=5"], _INTERNAL),
    ("op_launch_row6", CSRC / "op_launch_row.hip", ["-DOPL_ROW_PART=6"], _INTERNAL),
    ("op_launch_layer32", CSRC / "op_launch_layer32.hip", [], _INTERNAL),
    ("op_launch_attn", CSRC / "op_launch_attn.hip", [], _INTERNAL),
    ("op_launch_panel", CSRC / "op_launch_panel.hip", [], _INTERNAL),
]
SOURCES = sorted({u[1] for u in UNITS})
HEADERS = sorted({h for u in UNITS for h in u[3]})

FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wno-unused-result",
    # no SLP vectorizer: it fuses adjacent scalar fp32 ops of the epilogues into v_pk_* instructions, which cost
    # more issue time beside MFMAs than the two scalar ops they replace (measured: +1.1 % pairs/s without it)
    "-fno-slp-vectorize",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH and /opt/rocm/bin/hipcc)")


def _unit_digest(unit) -> str:
    _, src, defines, headers = unit
    digest = hashlib.sha256(" ".join(FLAGS + defines).encode())
    for path in [src, *headers]:
        digest.update(path.read_bytes())
    return digest.hexdigest()


def _unit_is_stale(unit) -> bool:
    obj = BUILD_DIR / f"{unit[0]}.o"
    stamp = BUILD_DIR / f"{unit[0]}.sha256"
    return not (obj.exists() and stamp.exists() and stamp.read_text() == _unit_digest(unit))


def is_stale() -> bool:
    if not OUTPUT.exists():
        return True
    built = OUTPUT.stat().st_mtime
    return any(p.stat().st_mtime > built for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile every HIP source into the shared library next to the package; returns its path."""

    if not force and not is_stale():
        return OUTPUT
    hipcc = _hipcc()
    BUILD_DIR.mkdir(parents=True, exist_ok=True)

    def compile_unit(unit) -> None:
        name, src, defines, _ = unit
        obj = BUILD_DIR / f"{name}.o"
        cmd = [hipcc, *FLAGS, *defines, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name} {defines} ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
        (BUILD_DIR / f"{name}.sha256").write_text(_unit_digest(unit))

    todo = [u for u in UNITS if force or _unit_is_stale(u)]
    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as pool:
        list(pool.map(compile_unit, todo))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUTPUT), *[str(BUILD_DIR / f"{u[0]}.o") for u in UNITS]]
    if verbose:
        print(" ".join(link), flush=True)
    proc = subprocess.run(link, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"link failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    return OUTPUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
