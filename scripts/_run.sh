python scripts/pair_kernel_debug.py 2>&1 | tail -6
python scripts/pair_kernel_check.py 2>&1 | tail -8
