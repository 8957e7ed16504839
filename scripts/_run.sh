bash scripts/rocprof_pass.sh r06_xsmall_refinit > gpurun_out/r06_pass.log 2>&1
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_style.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_style2.json 2>/dev/null
