#!/bin/bash
# Where a grid frame's loop kernel spends its time: a build of libmtfhip.so with -DMTFHIP_GRID_TRACE (wall-clock stamps of one
# workgroup of k_iclk_track) into scratch/ -- run this HERE, then on the GPU box: python tools/grid_trace.py
cd "$(dirname "$0")/../mtf_amd/csrc" || exit 1
make -s -j8 OUT=../../build/variants/libmtfhip_gtrace.so EXTRA="-DMTFHIP_GRID_TRACE" || exit 1
