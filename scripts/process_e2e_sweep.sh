# process() end to end over tokenizers / worker threads / request sizes (one JSON line each) -> gpurun_out/process_e2e.txt
OUT=gpurun_out/process_e2e.txt
: > $OUT
for N in 256 1024; do
  timeout 300 python scripts/process_e2e.py --contexts $N >> $OUT 2>/dev/null
  for W in 0 4; do
    timeout 300 python scripts/process_e2e.py --contexts $N --tokenizer wordpiece --workers $W >> $OUT 2>/dev/null
    TOKENIZERS_PARALLELISM=false timeout 300 python scripts/process_e2e.py --contexts $N --tokenizer wordpiece --workers $W | sed 's/^{/{"tokenizers_parallelism": "false", /' >> $OUT 2>/dev/null
  done
done
timeout 300 python scripts/process_e2e.py --contexts 256 --chars 400 --chars-are-words --tokenizer wordpiece --workers 0 >> $OUT 2>/dev/null
TOKENIZERS_PARALLELISM=false timeout 300 python scripts/process_e2e.py --contexts 256 --chars 400 --chars-are-words --tokenizer wordpiece --workers 0 | sed 's/^{/{"tokenizers_parallelism": "false", /' >> $OUT 2>/dev/null
cat $OUT
