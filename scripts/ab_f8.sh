# same-box A/B of the fp16 + e4m3 kernel sets against the (hi, lo) bf16 sets: W = bf16 | fp32 (checkpoint dtype)
W=${1:-bf16}
for i in 1 2 3; do
  OPEN_PROVENCE_NO_F8=1 python bench.py --steps 60 --no-cpu-baseline --no-long --no-base --weights $W > gpurun_out/ab_${W}_nof8_$i.json 2>/dev/null
  python bench.py --steps 60 --no-cpu-baseline --no-long --no-base --weights $W > gpurun_out/ab_${W}_f8_$i.json 2>/dev/null
done
python - $W <<'PY'
import json,glob,sys
W=sys.argv[1]
for tag in ("nof8","f8"):
    for f in sorted(glob.glob(f"gpurun_out/ab_{W}_{tag}_*.json")):
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d["roofline"]
        print(W, tag, round(d["value"]), "one:", round(d["one_pipeline"]["value"]), d["config"]["policy"]["kernel_set"], "dom", r["kernel"], round(r["avg_launch_ms"],4), "frac", round(r["frac"],4), {k:round(v,3) for k,v in d["kernel_ms_per_forward"].items()})
PY
