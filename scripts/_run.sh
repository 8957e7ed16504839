echo "# scripts/forward_fuzz.py --init refinit --models xsmall --trials 60 --flags NO_SMALL_BLOCKS: every ragged batch through the WAVE-PAIR whole-layer kernel (by default it runs from one 128-row block per CU upward); second run: the same batches on the 8 x 16 kernel (--flags NO_SMALL_BLOCKS,NO_LAYER_PAIRS)"
python scripts/forward_fuzz.py --init refinit --models xsmall --trials 60 --flags NO_SMALL_BLOCKS 2>&1 | grep -v amdgpu.ids
python scripts/forward_fuzz.py --init refinit --models xsmall --trials 60 --flags NO_SMALL_BLOCKS,NO_LAYER_PAIRS 2>&1 | grep -v amdgpu.ids
