"""Summarise the basic blocks of one kernel in the device ISA (hipcc -S --cuda-device-only output): instruction
mix per block and, with --pattern LABEL, the run-length pattern (M mfma, V valu, D lds, G global/buffer,
W s_waitcnt, B barrier, S scalar).  CPU-only tuning aid."""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
def main():
    key = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else None
    if not Path('/tmp/op.s').exists() or '--rebuild' in sys.argv:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-o", "/tmp/op.s", str(ROOT / "open_provence_amd/csrc/op_api.hip")], cwd="/tmp", capture_output=True)
    s = open('/tmp/op.s').read()
    names = [m.group(1) for m in re.finditer(r'^(_ZN3opk\w+):', s, re.M) if key in m.group(1)]
    name = names[0]
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    blocks = []; cur = ['entry', []]
    for l in s[i:j].splitlines():
        t = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            blocks.append(cur); cur = [m.group(1), []]
        elif t and not t.startswith(';') and not t.startswith('.'):
            cur[1].append(t)
    blocks.append(cur)
    def cls(x):
        op = x.split()[0]
        if 'mfma' in op: return 'M'
        if op.startswith('v_'): return 'V'
        if op.startswith('ds_'): return 'D'
        if op.startswith(('global_', 'buffer_', 'scratch_')): return 'G'
        if op.startswith('s_waitcnt'): return 'W'
        if op.startswith('s_barrier'): return 'B'
        return 'S'
    print(name)
    for lab, ins in blocks:
        c = [cls(x) for x in ins]
        print(f"{lab:10s} n={len(ins):4d} " + " ".join(f"{k}={c.count(k)}" for k in "MVDGWBS"))
        if pat and lab.endswith(pat):
            out = []; prev = None; n = 0
            for x, k in zip(ins, c):
                if k == 'W':
                    k = 'W[' + x.split(None, 1)[1] + ']'
                if k == prev: n += 1
                else:
                    if prev: out.append(f"{prev}{n if n > 1 else ''}")
                    prev, n = k, 1
            out.append(f"{prev}{n if n > 1 else ''}")
            print("   " + " ".join(out))
main()
