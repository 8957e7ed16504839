bash scripts/rocprof_pass.sh r06_xsmall_refinit > gpurun_out/r06_pass.log 2>&1; tail -3 gpurun_out/r06_pass.log | cut -c1-200
bash scripts/rocprof_pass.sh r06_base_refinit --model base > gpurun_out/r06_pass_base.log 2>&1; tail -3 gpurun_out/r06_pass_base.log | cut -c1-200
