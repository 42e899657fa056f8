#!/bin/bash
# Ablation of k_iclk_track's tolerance-mode loop: variants of libmtfhip.so with -DMTFHIP_GRID_ABL=<bits> (1 no texel fetch,
# 2 no workgroup reduction, 4 no solve) into scratch/ (run HERE); on the GPU box: MTFHIP_LIB=... python bench.py --workload grid
cd "$(dirname "$0")/../mtf_amd/csrc" || exit 1
for n in 1 2 4 7; do
  make -s -j8 OUT=../../scratch/libmtfhip_gabl$n.so EXTRA="-DMTFHIP_GRID_ABL=$n" || exit 1
done
