for N in 256 1024 4096; do
for W in 0 1 2; do
  timeout 300 python scripts/process_e2e.py --contexts $N --tokenizer wordpiece --workers $W 2>/dev/null | cut -c1-200
done
TOKENIZERS_PARALLELISM=false timeout 300 python scripts/process_e2e.py --contexts $N --tokenizer wordpiece --workers 4 2>/dev/null | cut -c1-200
done
