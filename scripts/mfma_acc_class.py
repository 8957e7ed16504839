"""Disassemble one compiled unit (build/hip/<unit>.o) and count, per kernel, the MFMAs whose accumulator is an AGPR
vs a VGPR.  On gfx950 a wave issuing MFMAs back to back runs them every 16 cycles with AGPR accumulators and every
24.5 with VGPR accumulators (microbench/mfma_loop.hip, clock probe) -- it matters for kernels at one wave per SIMD.
usage: mfma_acc_class.py <unit, e.g. op_launch_row3> [name filter]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
unit = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
obj = ROOT / "build" / "hip" / f"{unit}.o"
tmp = Path("/tmp/mfma_acc")
tmp.mkdir(exist_ok=True)
work = tmp / f"{unit}.o"
work.write_bytes(obj.read_bytes())
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(work)], check=True, capture_output=True)
device = next(tmp.glob(f"{unit}.o.*gfx950"))
asm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", str(device)], capture_output=True, text=True).stdout
cur, counts = None, {}
for line in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
    if m:
        cur = m.group(1)
        counts[cur] = {"a": 0, "v": 0, "accmov": 0, "insts": 0}
        continue
    if cur is None:
        continue
    t = line.strip().split()
    if not t:
        continue
    op = t[0]
    counts[cur]["insts"] += 1
    if op.startswith("v_mfma"):
        dst = t[1]
        counts[cur]["a" if dst.startswith("a") else "v"] += 1
    elif op.startswith("v_accvgpr"):
        counts[cur]["accmov"] += 1
names = subprocess.run(["c++filt"], input="\n".join(counts), capture_output=True, text=True).stdout.splitlines()
print(f"{'mfma->AGPR':>10} {'mfma->VGPR':>10} {'accvgpr mov':>11} {'insts':>7}  kernel")
for (k, c), name in zip(counts.items(), names):
    if c["a"] + c["v"] == 0 or flt not in name:
        continue
    print(f"{c['a']:10d} {c['v']:10d} {c['accmov']:11d} {c['insts']:7d}  {name[:110]}")
