#!/usr/bin/env python
"""Resolve a set of compile-time measurement switches in the kernel headers as NOT defined (a small `unifdef`).

The ablation hooks of the whole-layer kernel (OPK_ABL_*, OPK_QKV_*, OPK_PLAIN_*) are measurement tools of
microbench/rowgemm_ablate.hip; the library never defines them.  They live as a patch under microbench/experiments/
(`rowgemm_ablation_hooks.patch`, applied to a scratch copy of csrc/ by scripts/ablate_x.sh / ablate_rowgemm.sh); this
script is what produced the hook-free product headers from the instrumented ones, and checks that a header has none.

    python scripts/strip_switches.py --check                 # exit 1 if a product header still names a stripped switch
    python scripts/strip_switches.py FILE... [--in-place]    # resolve the switches in FILE (stdout, or in place)
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "open_provence_amd" / "csrc"

# name -> value the product build sees (None: not defined)
SWITCHES: dict[str, int | None] = {
    "OPK_ABL_NO_BARRIER": None,
    "OPK_ABL_NO_DMA": None,
    "OPK_ABL_NO_FRAG_READS": None,
    "OPK_ABL_NO_FRAG_WAIT": None,
    "OPK_ABL_NO_MLP_VALU": None,
    "OPK_ABL_NO_QKV_RIDERS": None,
    "OPK_ABL_NO_QKV_STORE": None,
    "OPK_QKV_SINGLE": None,
    "OPK_QKV_STREAM_TIMING": None,
    "OPK_PLAIN_LOADS": None,
    "OPK_PLAIN_STORES": None,
}

_COND = re.compile(r"^\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$")


def _evaluate(kind: str, rest: str) -> bool | None:
    """Truth of a conditional that only names stripped switches; None = leave the directive alone."""

    rest = rest.split("//")[0].strip()
    if kind in ("ifdef", "ifndef"):
        if rest not in SWITCHES:
            return None
        defined = SWITCHES[rest] is not None
        return defined if kind == "ifdef" else not defined
    if kind == "if":
        names = re.findall(r"defined\((\w+)\)", rest)
        if names and all(n in SWITCHES for n in names) and re.fullmatch(r"(defined\(\w+\)\s*(\|\||&&)?\s*)+", rest):
            expr = re.sub(r"defined\((\w+)\)", lambda m: str(SWITCHES[m.group(1)] is not None), rest)
            return bool(eval(expr.replace("||", " or ").replace("&&", " and ")))  # noqa: S307 - our own constants
    return None


def strip(text: str) -> str:
    out: list[str] = []
    # stack of (resolved?, emitting, parent_emitting); resolved blocks drop their directives
    stack: list[list] = []
    emitting = True
    for line in text.splitlines(keepends=True):
        m = _COND.match(line)
        if not m:
            if emitting:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("ifdef", "ifndef", "if"):
            truth = _evaluate(kind, rest) if emitting else None
            stack.append([truth is not None, truth, emitting])
            if truth is None:
                if emitting:
                    out.append(line)
            else:
                emitting = emitting and truth
        elif kind in ("else", "elif"):
            resolved, truth, parent = stack[-1]
            if resolved:
                if kind == "elif":
                    raise ValueError("#elif on a stripped switch is not handled")
                stack[-1][1] = not truth
                emitting = parent and not truth
            elif emitting:
                out.append(line)
        else:  # endif
            resolved, _truth, parent = stack.pop()
            if resolved:
                emitting = parent
            elif emitting:
                out.append(line)
    if stack:
        raise ValueError("unbalanced conditionals")
    return "".join(out)


def main() -> None:
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--check" in sys.argv:
        bad = [(p.name, n) for p in sorted(CSRC.glob("*")) for n in SWITCHES
               if re.search(rf"#\s*(ifdef|ifndef|if)\b[^\n]*\b{n}\b", p.read_text())]
        for name, switch in bad:
            print(f"{name}: conditional on {switch}")
        sys.exit(1 if bad else 0)
    for name in args:
        path = Path(name)
        result = strip(path.read_text())
        if "--in-place" in sys.argv:
            path.write_text(result)
        else:
            sys.stdout.write(result)


if __name__ == "__main__":
    main()
