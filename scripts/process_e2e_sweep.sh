# process() end to end over tokenizers / worker threads / request sizes (one JSON line each) -> gpurun_out/process_e2e.txt
# default: TOKENIZERS_PARALLELISM=false (as the reference) and, for a fast tokenizer from 128 jobs on, 4 worker threads
OUT=gpurun_out/process_e2e.txt
: > $OUT
for N in 256 1024 4096; do
  timeout 300 python scripts/process_e2e.py --contexts $N >> $OUT 2>/dev/null
  timeout 300 python scripts/process_e2e.py --contexts $N --tokenizer wordpiece | sed 's/^{/{"config": "default", /' >> $OUT 2>/dev/null
  for W in 0 2 8; do
    timeout 300 python scripts/process_e2e.py --contexts $N --tokenizer wordpiece --workers $W >> $OUT 2>/dev/null
  done
  TOKENIZERS_PARALLELISM=true timeout 300 python scripts/process_e2e.py --contexts $N --tokenizer wordpiece --workers 0 | sed 's/^{/{"tokenizers_parallelism": "true", /' >> $OUT 2>/dev/null
done
timeout 300 python scripts/process_e2e.py --contexts 256 --chars 400 --chars-are-words --tokenizer wordpiece | sed 's/^{/{"config": "default", /' >> $OUT 2>/dev/null
cat $OUT
