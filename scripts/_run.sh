echo "# round 6: process() with DEFAULT arguments after the stream change (two unpartitioned launch sequences) and the ungated wave-pair kernel"
for args in "--contexts 4096 --reps 8" "--contexts 2048 --reps 8" "--contexts 1024 --reps 5" "--contexts 256 --reps 5"; do
echo "\$ python scripts/process_e2e.py --tokenizer wordpiece --stock-tokenizer $args"
python scripts/process_e2e.py --tokenizer wordpiece --stock-tokenizer $args 2>&1 | grep "^{" | tail -1
done
