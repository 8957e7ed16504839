#!/bin/bash
# Build (CPU, cross-compiled) or run (GPU box) experiment variants of the whole-layer kernel through
# microbench/rowgemm_ablate.hip.  Variants are listed in scripts/ablate_x.list: "<name> <F8: 1|2> <defines...>".
#   scripts/ablate_x.sh build | run [rounds]
set -u
cd "$(dirname "$0")/.."
LIST=scripts/ablate_x.list
if [ "${1:-run}" = build ]; then
  rm -f microbench/ablx_*
  CSRC=$(bash scripts/instrumented_csrc.sh)  # csrc/ + the ablation hooks (microbench/experiments/rowgemm_ablation_hooks.patch)
  while read -r NAME F8 DEFS; do
    [ -z "$NAME" ] && continue
    case $NAME in \#*) continue;; esac
    T=1; [ "$F8" = 2 ] && T=3
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I $CSRC -DOPK_TIMING -DABL_F8=$F8 -DABL_T=$T -DABL_LAYER=2 $DEFS \
      -DABL_NAME="\"$NAME\"" -o microbench/ablx_$NAME microbench/rowgemm_ablate.hip &
    while [ "$(jobs -r | wc -l)" -ge 6 ]; do sleep 1; done
  done < $LIST
  wait
  ls microbench/ablx_* | wc -l
else
  for round in $(seq 1 ${2:-2}); do
    while read -r NAME F8 DEFS; do
      [ -z "$NAME" ] && continue
      case $NAME in \#*) continue;; esac
      timeout 120 ./microbench/ablx_$NAME
    done < $LIST
  done
fi
