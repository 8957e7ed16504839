python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_suite.txt 2>&1; grep -n "passed\|failed" gpurun_out/r06_gpu_suite.txt | tail -3
