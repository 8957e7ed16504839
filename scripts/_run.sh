python -m pytest tests/test_gpu_calibration.py -x -q -m gpu -k "mlp_correction or panel_path_calibrates or deep_panel" 2>&1 | tail -15
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-long --no-trained-like 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d['base_model']
        for k in ('fp32_checkpoint','bf16_checkpoint'):
            v=b[k]; print(k, v['value'], v['ms_per_step'], v['kernel_set'], v['calibration'].get('mlp_correction_layers'), v['calibration'].get('mlp_correction_err'), v['calibration']['audit'])
"
