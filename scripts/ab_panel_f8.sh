# same-box A/B of the fp16 + e4m3 kernel sets on the panel path (hidden 512 / 768): M = base | en-gte | large, W = bf16 | fp32
M=${1:-base}
W=${2:-bf16}
S=${3:-20}
for i in 1 2; do
  timeout 600 python bench.py --model $M --steps $S --no-cpu-baseline --no-long --no-other-dtype --weights $W > gpurun_out/abp_${M}_${W}_nof8_$i.json 2>gpurun_out/abp_err.log
  OPEN_PROVENCE_PANEL_F8=1 timeout 600 python bench.py --model $M --steps $S --no-cpu-baseline --no-long --no-other-dtype --weights $W > gpurun_out/abp_${M}_${W}_f8_$i.json 2>>gpurun_out/abp_err.log
done
python - $M $W <<'PY'
import json,glob,sys
M,W=sys.argv[1:3]
for tag in ("nof8","f8"):
    for f in sorted(glob.glob(f"gpurun_out/abp_{M}_{W}_{tag}_*.json")):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "unreadable", e); continue
        r=d["roofline"]
        print(M, W, tag, round(d["value"]), d["config"]["policy"]["kernel_set"], "dom", r["kernel"], round(r["avg_launch_ms"],4), "frac", round(r["frac"],4), {k:round(v,3) for k,v in d["kernel_ms_per_forward"].items()})
PY
