#!/bin/bash
# Ablation of k_mi_pass_grad_hess: builds variants of libmtfhip.so with -DMTFHIP_MI_ABL=n into scratch/ (run HERE, CPU),
# then on the GPU box: MTFHIP_LIB=scratch/libmtfhip_abl<n>.so python bench.py --workload mi ...
cd "$(dirname "$0")/../mtf_amd/csrc" || exit 1
for n in 1 2; do
  make -s -j8 OUT=../../scratch/libmtfhip_abl$n.so EXTRA="-DMTFHIP_MI_ABL=$n" || exit 1
done
