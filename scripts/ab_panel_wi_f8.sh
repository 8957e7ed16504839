for i in 1 2; do
  for W in fp32 bf16; do
    python bench.py --model base --steps 30 --no-cpu-baseline --no-long --no-base --no-other-dtype --weights $W 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', '$W', round(d['value']), d['config']['policy']['kernel_set'], {k:round(v,2) for k,v in d['kernel_ms_per_forward'].items() if 'gemm' in k or 'norm' in k})"
    OPEN_PROVENCE_PANEL_F8_WI=1 python bench.py --model base --steps 30 --no-cpu-baseline --no-long --no-base --no-other-dtype --weights $W 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wi-f8  ', '$W', round(d['value']), d['config']['policy']['kernel_set'], {k:round(v,2) for k,v in d['kernel_ms_per_forward'].items() if 'gemm' in k or 'norm' in k})"
  done
done
