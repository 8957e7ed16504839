"""Per basic block of one kernel in a compiled unit: instruction counts by class (MFMA, VALU, accvgpr moves, LDS, VMEM,
SALU, waitcnt, nops) -- to see what a hot loop issues per MFMA.  usage: isa_blocks.py <unit> <kernel name substring> [min_insts]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
unit, pattern = sys.argv[1], sys.argv[2]
min_insts = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tmp = Path("/tmp/mfma_acc")
tmp.mkdir(exist_ok=True)
work = tmp / f"{Path(unit).stem}.o"
src_obj = Path(unit) if unit.endswith(".o") else ROOT / "build" / "hip" / f"{unit}.o"  # a unit name or an object path
work.write_bytes(src_obj.read_bytes())
unit = Path(unit).stem
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(work)], check=True, capture_output=True)
device = next(tmp.glob(f"{unit}.o.*gfx950"))
asm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--demangle", str(device)], capture_output=True, text=True).stdout
lines = asm.splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*>:$", l) and pattern in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <.*>:$", lines[i])), len(lines))
body = []
for l in lines[start + 1:end]:
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
    if m:
        body.append((int(m.group(3), 16), m.group(1), m.group(2)))
addr_index = {a: i for i, (a, _, _) in enumerate(body)}
# branch targets: s_cbranch / s_branch with simm16 -> target = addr + 4 + simm*4
leaders = {0}
edges = []
for i, (a, op, args) in enumerate(body):
    if op.startswith("s_cbranch") or op == "s_branch":
        m = re.search(r"(-?\d+)", args)
        if m:
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + off * 4
            if tgt in addr_index:
                leaders.add(addr_index[tgt])
                edges.append((i, addr_index[tgt]))
        leaders.add(i + 1)
leaders = sorted(x for x in leaders if x < len(body))
def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accmov"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op == "s_waitcnt": return "wait"
    if op == "s_nop": return "nop"
    if op == "s_barrier": return "barrier"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return "other"
print(f"kernel: {lines[start][:150]}\n{len(body)} instructions")
print(f"{'block@':>8} {'insts':>6} {'mfma':>5} {'valu':>5} {'accmov':>6} {'lds':>4} {'vmem':>5} {'salu':>5} {'wait':>5} {'nop':>4} {'bar':>4}  back-edge")
for bi, s in enumerate(leaders):
    e = leaders[bi + 1] if bi + 1 < len(leaders) else len(body)
    c = {}
    for _, op, _ in body[s:e]:
        c[klass(op)] = c.get(klass(op), 0) + 1
    n = e - s
    if n < min_insts:
        continue
    back = [f"->{body[t][0]:#x}" for (f, t) in edges if s <= f < e and t <= f]
    print(f"{body[s][0]:#8x} {n:6d} {c.get('mfma',0):5d} {c.get('valu',0):5d} {c.get('accmov',0):6d} {c.get('lds',0):4d} {c.get('vmem',0):5d} "
          f"{c.get('salu',0):5d} {c.get('wait',0):5d} {c.get('nop',0):4d} {c.get('barrier',0):4d}  {' '.join(back)}")

# optional: print one block as a class string (M mfma, v valu, a accvgpr move, L lds, G vmem, s salu, W waitcnt, n nop, B barrier)
if len(sys.argv) > 4:
    want = int(sys.argv[4], 16)
    sym = {"mfma": "M", "valu": "v", "accmov": "a", "lds": "L", "vmem": "G", "salu": "s", "wait": "W", "nop": "n", "barrier": "B", "other": "?"}
    for bi, s in enumerate(leaders):
        if body[s][0] != want:
            continue
        e = leaders[bi + 1] if bi + 1 < len(leaders) else len(body)
        text = "".join(sym[klass(op)] for _, op, _ in body[s:e])
        for i in range(0, len(text), 120):
            print(text[i:i + 120])
        if len(sys.argv) > 5:
            for a, op, args in body[s:e]:
                if klass(op) in ("wait", "nop"):
                    print(f"{a:#x} {op} {args}")

# optional: opcode histogram of one block: isa_blocks.py <unit> <kernel> <min> <block addr> hist
if len(sys.argv) > 5 and sys.argv[5] == "hist":
    want = int(sys.argv[4], 16)
    for bi, s in enumerate(leaders):
        if body[s][0] != want:
            continue
        e = leaders[bi + 1] if bi + 1 < len(leaders) else len(body)
        h = {}
        for _, op, _ in body[s:e]:
            h[op] = h.get(op, 0) + 1
        for op, n in sorted(h.items(), key=lambda kv: -kv[1]):
            print(f"{n:5d} {op}")
