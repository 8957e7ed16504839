set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; tail -c 400 gpurun_out/r06_bench_default.json
: > gpurun_out/r06_bench_configs.jsonl
python bench.py --model base --steps 100 --warmup 5 --no-cpu-baseline --no-long --no-base --no-trained-like --no-worst-case --no-other-dtype >> gpurun_out/r06_bench_configs.jsonl 2>/dev/null
python bench.py --model large --pairs 64 --seq-len 2048 --steps 100 --warmup 5 --no-cpu-baseline --no-long --no-base --no-trained-like --no-worst-case --no-other-dtype >> gpurun_out/r06_bench_configs.jsonl 2>/dev/null
python bench.py --model en-gte --varlen --steps 100 --warmup 5 --no-cpu-baseline --no-long --no-base --no-trained-like --no-worst-case --no-other-dtype >> gpurun_out/r06_bench_configs.jsonl 2>/dev/null
wc -l gpurun_out/r06_bench_configs.jsonl
python scripts/trained_like_probe.py > gpurun_out/r06_trained_like_probe.txt 2>&1; tail -5 gpurun_out/r06_trained_like_probe.txt
