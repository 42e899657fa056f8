#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages per launch."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(root + "/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-48:]
        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
kernels = sorted({k[0] for k in agg})
for kn in kernels:
    if not any(s in kn for s in ("fused", "score", "finish_track", "ncc", "mi_", "pf_", "iclk", "persist", "k_nn_", "template_init", "grid_fb")):
        continue
    print(kn)
    for (k, c), v in sorted(agg.items()):
        if k == kn:
            print("   %-36s n=%-4d avg=%.6g" % (c, len(v), sum(v) / len(v)))
