#!/usr/bin/env python
"""Fingerprint of the row-path kernels' ISA: per kernel the instruction count, register counts, scratch, and a hash of the
mnemonic sequence (register names ignored).  Usage: isa_fp.py <csrc dir> <out json>"""
import hashlib, json, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

csrc = Path(sys.argv[1]); out = Path(sys.argv[2])
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --cuda-device-only -S".split()
tmp = Path(tempfile.mkdtemp())

def unit(part):
    dst = tmp / f"row{part}.s"
    subprocess.run(["hipcc", *flags, f"-DOPL_ROW_PART={part}", "-o", str(dst), str(csrc / "op_launch_row.hip")], check=True, capture_output=True)
    return dst.read_text()

with ThreadPoolExecutor(7) as pool:
    texts = list(pool.map(unit, range(7)))
fp = {}
for t in texts:
    for m in re.finditer(r"^(_ZN3opk[^:\s]+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", t, re.S | re.M):
        name, body = m.group(1), m.group(2)
        ops = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        fp[name] = {"n": len(ops), "hash": hashlib.sha1(" ".join(ops).encode()).hexdigest()[:12]}
    for m in re.finditer(r"\.name:\s+(_ZN3opk\S+)\n(.*?)\.wavefront_size", t, re.S):
        name = m.group(1)
        meta = m.group(2)
        d = fp.setdefault(name, {})
        for key in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count"):
            mm = re.search(rf"\.{key}:\s+(\d+)", meta)
            if mm: d[key] = int(mm.group(1))
out.write_text(json.dumps(fp, indent=1, sort_keys=True))
print(len(fp), "kernels")
