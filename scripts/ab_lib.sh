# same-box A/B of two builds of the library: scripts/ab_lib.sh <other .so> [bench args...]  (three alternations)
LIB=$1; shift
for i in 1 2 3; do
  OPEN_PROVENCE_HIP_LIB=$LIB OPEN_PROVENCE_HIP_LIB_ANY_ABI=1 python bench.py --steps 60 --no-cpu-baseline --no-long --no-base "$@" > gpurun_out/ablib_other_$i.json 2>/dev/null
  python bench.py --steps 60 --no-cpu-baseline --no-long --no-base "$@" > gpurun_out/ablib_tree_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for tag in ("other","tree"):
    for f in sorted(glob.glob(f"gpurun_out/ablib_{tag}_*.json")):
        d=json.loads(open(f).read().strip().splitlines()[-1])
        b=d.get("bf16_checkpoint",{})
        print(tag, round(d["value"]), "bf16", round(b.get("value",0)), {k:round(v,3) for k,v in d["kernel_ms_per_forward"].items() if k.startswith(("attn","fused"))})
PY
