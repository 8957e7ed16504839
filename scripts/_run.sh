bash scripts/_mb.sh p16_buf p16_f16_xt 2>&1 | grep "cycles\|avg\|determ"
export OPEN_PROVENCE_WRITE_ORACLE_CACHE=$PWD/gpurun_out/oracle_cache
rm -rf $OPEN_PROVENCE_WRITE_ORACLE_CACHE; mkdir -p $OPEN_PROVENCE_WRITE_ORACLE_CACHE
timeout 2600 python -m pytest tests/test_gpu_timed_path.py tests/test_gpu_calibration.py tests/test_gpu_decisions.py -m gpu -x -q 2>&1 | tail -4
ls gpurun_out/oracle_cache | wc -l
