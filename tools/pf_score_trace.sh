#!/bin/bash
# Where a small scoring launch spends its time: a build of libmtfhip.so with -DMTFHIP_PF_TRACE (wall-clock stamps of workgroup 0 of
# k_pf_score) into scratch/ -- run this HERE, then on the GPU box: MTFHIP_LIB=build/variants/libmtfhip_pftrace.so python tools/pf_score_trace.py [n]
cd "$(dirname "$0")/../mtf_amd/csrc" || exit 1
make -s -j8 OUT=../../build/variants/libmtfhip_pftrace.so EXTRA="-DMTFHIP_PF_TRACE" || exit 1
