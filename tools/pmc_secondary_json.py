#!/usr/bin/env python
"""Per-kernel averages per launch of the secondary workloads' PMC passes (tools/r05_final.sh: <root>/pmc_mi, pmc_pf, pmc_grid, one
directory per rocprofv3 --pmc pass) as profiles/pmc_secondary_latest.json -- what bench.py's grid / pf / mi lines quote as `traffic`
and as VALU issue, guarded by the hash of the kernel sources the counters were collected on.   usage: pmc_secondary_json.py <root> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

root, out_path = sys.argv[1], sys.argv[2]
KEEP = ("k_mi_pass_hist", "k_mi_pass_grad_hess", "k_mi_tables_iter", "k_mi_finish_fast", "k_pf_score", "k_pf_scan", "k_pf_select", "k_pf_iter", "k_iclk_track", "k_template_init", "k_mi_tables_poly", "k_nn_dataset")
out = {"kernel_sources_sha": bench.kernel_sources_sha(), "commit": open(".git_head").read().strip() if os.path.exists(".git_head") else None,
       "correction": "HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (counters in KiB; gfx950 FETCH_SIZE half-count, MI355X_MICROARCH.md HBM section)"}
for wl in ("mi", "pf", "grid", "nn"):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "pmc_" + wl, "p*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].split(" ")[-1]
            if name in KEEP:
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    sec = {}
    for (k, c), v in sorted(agg.items()):
        sec.setdefault(k, {})[c] = sum(v) / len(v)
        sec[k]["launches_counted"] = len(v)
    for k, d in sec.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            d["traffic_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
    out[wl] = sec
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: sorted(v) for k, v in out.items() if isinstance(v, dict)}))
