#!/bin/bash
# Same-box A/B of several builds of the library on the whole forward: scripts/ab_multi.sh "<lib1> <lib2> ..." [bench args]
# ("tree" = the library in the tree); three alternations, prints value + the whole-layer kernel's time per forward.
LIBS=$1; shift
mkdir -p gpurun_out/abm
for i in 1 2 3; do
  for L in $LIBS; do
    if [ "$L" = tree ]; then
      python bench.py --steps 60 --no-cpu-baseline --no-long --no-base --no-other-dtype "$@" > gpurun_out/abm/tree_$i.json 2>/dev/null
    else
      OPEN_PROVENCE_HIP_LIB=ab_libs/$L.so OPEN_PROVENCE_HIP_LIB_ANY_ABI=1 python bench.py --steps 60 --no-cpu-baseline --no-long --no-base --no-other-dtype "$@" > gpurun_out/abm/${L}_$i.json 2>/dev/null
    fi
  done
done
python - $LIBS <<'PY'
import json,glob,sys
for L in sys.argv[1:]:
    vals=[]
    for f in sorted(glob.glob(f"gpurun_out/abm/{L}_*.json")):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:
            print(L, f, "unreadable"); continue
        k=d["kernel_ms_per_forward"]
        vals.append(d["value"])
        print(f"{L:14s} {d['value']:9.0f} pairs/s  layer {k.get('fused_layer_attnout_mlp_qkv',0):.3f} ms  attn {k.get('attn_global',0)+k.get('attn_local',0):.3f}  clock {d.get('shader_clock_ghz',{}).get('value',0):.3f}")
    if vals: print(f"{L:14s} mean {sum(vals)/len(vals):9.0f}")
PY
