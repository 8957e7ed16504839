# same-box A/B of two library builds on a panel-path model: scripts/ab_panel_lib.sh <other .so> <model> <bf16|fp32> [steps]
LIB=$1; M=${2:-base}; W=${3:-bf16}; S=${4:-20}
for i in 1 2; do
  OPEN_PROVENCE_HIP_LIB=$LIB timeout 600 python bench.py --model $M --steps $S --no-cpu-baseline --no-long --no-base --no-other-dtype --weights $W > gpurun_out/abpl_other_$i.json 2>gpurun_out/abpl_err.log
  timeout 600 python bench.py --model $M --steps $S --no-cpu-baseline --no-long --no-base --no-other-dtype --weights $W > gpurun_out/abpl_tree_$i.json 2>>gpurun_out/abpl_err.log
done
python - $M $W <<'PY'
import json,glob,sys
M,W=sys.argv[1:3]
for tag in ("other","tree"):
    for f in sorted(glob.glob(f"gpurun_out/abpl_{tag}_*.json")):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "unreadable", e); continue
        r=d["roofline"]
        print(M, W, tag, round(d["value"]), d["config"]["policy"]["kernel_set"], "dom", r["kernel"], round(r["avg_launch_ms"],4), {k:round(v,3) for k,v in d["kernel_ms_per_forward"].items() if k.startswith(("gemm","attn","layer"))})
PY
