# same-box A/B of one environment switch: scripts/ab_env.sh VAR [bench args...]  (three alternations, default bench)
VAR=$1; shift
for i in 1 2 3; do
  python bench.py --steps 60 --no-cpu-baseline --no-long --no-base "$@" > gpurun_out/abenv_off_$i.json 2>/dev/null
  env $VAR=1 python bench.py --steps 60 --no-cpu-baseline --no-long --no-base "$@" > gpurun_out/abenv_on_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for tag in ("off","on"):
    for f in sorted(glob.glob(f"gpurun_out/abenv_{tag}_*.json")):
        d=json.loads(open(f).read().strip().splitlines()[-1])
        b=d.get("bf16_checkpoint",{})
        print(tag, round(d["value"]), "bf16", round(b.get("value",0)), {k:round(v,3) for k,v in d["kernel_ms_per_forward"].items() if k.startswith(("attn","fused"))})
PY
