#!/bin/bash
# Scratch copy of the kernel headers WITH the ablation hooks: build/ablate_csrc/ = open_provence_amd/csrc/ +
# microbench/experiments/rowgemm_ablation_hooks.patch.  The product headers carry no ablation switches
# (scripts/strip_switches.py --check); the microbenchmarks that need them compile against this copy (-I).
# Prints the directory.
set -eu
cd "$(dirname "$0")/.."
DST=build/ablate_csrc
rm -rf $DST && mkdir -p $DST/open_provence_amd
cp -r open_provence_amd/csrc $DST/open_provence_amd/
(cd $DST && patch -p1 -s < ../../microbench/experiments/rowgemm_ablation_hooks.patch)
echo "$PWD/$DST/open_provence_amd/csrc"
