for r in 1 2 3; do for pp in 1 2; do
python bench.py --init o1 --pipelines $pp --steps 60 --warmup 5 --no-cpu-baseline --no-long --no-base --no-trained-like --no-worst-case --no-other-dtype 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('o1 fp32 pipelines $pp', round(d['value']), d['ms_per_step'], d['config']['policy']['kernel_set'], d['config']['output_checksum']['stored'])
"
python bench.py --init o1 --weights bf16 --pipelines $pp --steps 60 --warmup 5 --no-cpu-baseline --no-long --no-base --no-trained-like --no-worst-case --no-other-dtype 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('o1 bf16 pipelines $pp', round(d['value']), d['ms_per_step'], d['config']['policy']['kernel_set'], d['config']['output_checksum']['stored'])
"
done; done
