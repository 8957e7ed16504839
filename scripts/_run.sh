for r in 1 2 3 4; do for g in -1 1 0; do
OPEN_PROVENCE_PIPELINE_MASK_GROUP=$g python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-long --no-base --no-trained-like --no-worst-case --no-other-dtype 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('group $g', round(d['value']), round(d['one_pipeline']['value']), round(d['step_ms']['p10'],3), round(d['step_ms']['median'],3), round(d['step_ms']['p90'],3))
"
done; done
