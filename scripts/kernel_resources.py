"""Compile the HIP library with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, AGPRs, spills, scratch, occupancy, LDS).  CPU-only; used while tuning register pressure."""

from __future__ import annotations

import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "open_provence_amd" / "csrc"


def main() -> None:
    """usage: kernel_resources.py <unit: api|row0|row1|row2|attn|panel> [name filter]"""

    unit = sys.argv[1] if len(sys.argv) > 1 else "attn"
    pattern = sys.argv[2] if len(sys.argv) > 2 else ""
    src, defines = {"api": ("op_api.hip", []), "attn": ("op_launch_attn.hip", []), "panel": ("op_launch_panel.hip", []),
                    "row0": ("op_launch_row.hip", ["-DOPL_ROW_PART=0"]), "row1": ("op_launch_row.hip", ["-DOPL_ROW_PART=1"]),
                    "row2": ("op_launch_row.hip", ["-DOPL_ROW_PART=2"]), "row3": ("op_launch_row.hip", ["-DOPL_ROW_PART=3"])}[unit]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-c", *defines,
           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/op_resources.o", str(CSRC / src)]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            cur = {"name": text.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in text:
            k, v = text.split(":", 1)
            cur[k.strip()] = v.strip()
    demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                              capture_output=True, text=True).stdout.splitlines()
    print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>5} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
    for r, name in zip(rows, demangle):
        name = re.sub(r"\(.*\)$", "", name).replace("void opk::", "")
        if pattern and pattern not in name:
            continue
        print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('VGPRs Spill','?'):>5} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('Occupancy [waves/SIMD]','?'):>3} "
              f"{r.get('LDS Size [bytes/block]','?'):>6}  {name}")


if __name__ == "__main__":
    main()
