for r in 1 2; do for v in 0 1; do
OPK_ATTN_LOCAL_WAVES=$((4+4*v)) python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-long --no-base --no-trained-like 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('LW=$v', round(d['value']), d['one_pipeline']['value'], d['kernel_ms_per_forward'])
"
done; done
