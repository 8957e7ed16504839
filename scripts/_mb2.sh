for off in 0 10 20 35 50 70; do ABL_ROWS=65536 ABL_DUAL=$off timeout 120 ./microbench/ablx_p16_f16_xt 2>&1 | grep "dual\|cycles per" | cut -c1-230; done
echo interleaved; for off in 0 35; do ABL_DUAL_INTERLEAVE=1 ABL_ROWS=65536 ABL_DUAL=$off timeout 120 ./microbench/ablx_p16_f16_xt 2>&1 | grep "dual\|cycles per" | cut -c1-230; done
