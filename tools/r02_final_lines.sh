#!/bin/bash
# The bench.py lines of the round's final code, one gpurun call: gpurun_out/r02_final_bench_lines.jsonl
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r02_final_bench_lines.jsonl; : > $out
timeout 600 python bench.py 2>/dev/null | tail -1 >> $out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 >> $out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 >> $out
for wl in grid pf mi; do
  steps=200; [ $wl = mi ] && steps=5
  timeout 600 python bench.py --workload $wl --steps $steps --warmup 5 --cpu-seconds 4 2>/dev/null | tail -1 >> $out
done
timeout 300 python bench.py --workload dropin --sm esm --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $out
timeout 300 python bench.py --workload dropin --sm esm --device-loop --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 >> $out
python - <<'PY'
import json
for l in open("gpurun_out/r02_final_bench_lines.jsonl"):
    d = json.loads(l); r = d.get("roofline") or {}
    print("%-70s %12.0f %9.2f us frac=%s lean=%s" % (d["metric"][:70], d["value"], d["ms_per_step"] * 1e3, r.get("frac"), (d.get("lean") or {}).get("value")))
PY
