#!/usr/bin/env python
"""Compare two ISA fingerprints written by scripts/isa_fingerprint.py: which instantiations changed in mnemonic sequence, instruction
count, register counts or scratch.  Usage: isa_compare.py before.json after.json"""
import json, sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
same=0
for k in sorted(a):
    x,y=a[k],b.get(k)
    if x==y: same+=1; continue
    tag = 'SAME-OPS' if y and x.get('hash')==y.get('hash') else 'ops differ'
    print(k[23:-22], tag, 'n', x.get('n'), '->', (y or {}).get('n'), '| vgpr', x.get('vgpr_count'), '->', (y or {}).get('vgpr_count'), '| agpr', x.get('agpr_count'), '->', (y or {}).get('agpr_count'), '| scratch', x.get('private_segment_fixed_size'), '->', (y or {}).get('private_segment_fixed_size'))
print(same, 'of', len(a), 'kernels identical in mnemonics and register counts;', len(set(b)-set(a)), 'new')
