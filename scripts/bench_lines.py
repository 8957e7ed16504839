import json,sys
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("{"):
            d=json.loads(line); r=d["roofline"]
            print(path.split("/")[-1], round(d["value"]), "pairs/s", {k: round(v,2) for k,v in d["kernel_ms_per_forward"].items() if v>0.5})
