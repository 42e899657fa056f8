#!/bin/bash
# Ablation of finish_track_body (the device-side solve + update of an iteration): variants of libmtfhip.so with
# -DMTFHIP_FIN_ABL=<bits> (1 no Gauss-Jordan elimination, 2 no compositional update / corner test, 4 operand loads only) into
# scratch/ (run HERE); on the GPU box: MTFHIP_LIB=... rocprofv3 --kernel-trace --stats -- python bench.py --mode lean ...
cd "$(dirname "$0")/../mtf_amd/csrc" || exit 1
for n in 1 2 3 4; do
  make -s -j8 OUT=../../scratch/libmtfhip_fabl$n.so EXTRA="-DMTFHIP_FIN_ABL=$n" || exit 1
done
