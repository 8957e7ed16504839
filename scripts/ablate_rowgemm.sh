#!/bin/bash
# Build (here, CPU) or run (GPU box) the ablation binaries of microbench/rowgemm_ablate.hip.
#   scripts/ablate_rowgemm.sh build   -> microbench/abl_*   (cross-compiled; they travel with the snapshot)
#   scripts/ablate_rowgemm.sh run     -> prints one line per binary, three rounds interleaved
set -u
cd "$(dirname "$0")/.."
VARIANTS="baseline NO_BARRIER NO_DMA NO_MLP_VALU"  # the switches rowgemm_ablation_hooks.patch still carries
if [ "${1:-run}" = build ]; then
  CSRC=$(bash scripts/instrumented_csrc.sh)
  for T in 1 3; do
    for V in $VARIANTS; do
      DEF=""; [ $V != baseline ] && DEF="-DOPK_ABL_$V"
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I $CSRC $DEF -DABL_T=$T -DABL_NAME="\"$V\"" \
        -o microbench/abl_${V}_t$T microbench/rowgemm_ablate.hip &
    done
    wait
  done
  # whole-layer kernel (ABL_LAYER = 1: no q/k/v loop, 2: with it), terms = 1 only
  for V in baseline NO_BARRIER NO_MLP_VALU NO_DMA; do
    for LY in 1 2; do
      DEF=""; [ $V != baseline ] && DEF="-DOPK_ABL_$V"
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I $CSRC $DEF -DABL_T=1 -DABL_LAYER=$LY -DABL_NAME="\"layer$LY-$V\"" \
        -o microbench/abl_layer${LY}_${V} microbench/rowgemm_ablate.hip &
    done
  done
  wait
  ls microbench/abl_* | wc -l
elif [ "${1:-run}" = layer ]; then
  for round in 1 2; do for f in microbench/abl_layer*; do ./$f; done; done
else
  for round in 1 2; do
    for T in 1 3; do for V in $VARIANTS; do ./microbench/abl_${V}_t$T; done; done
  done
fi
