#!/bin/bash
# The configurations tabulated in DESIGN.md section 5, one bench.py line each (one gpurun call): gpurun_out/r02_config_table.jsonl
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r02_config_table.jsonl; : > $out
run() { timeout 300 python bench.py "$@" --no-cpu 2>/dev/null | tail -1 >> $out; }
run --steps 200 --warmup 20
run --steps 200 --warmup 20 --targets 256 --no-lean
run --steps 200 --warmup 20 --mode lean --no-lean
run --steps 200 --warmup 20 --mode lean --math replay --no-lean
run --steps 200 --warmup 20 --sm fclk --no-lean
run --steps 200 --warmup 20 --sm iclk --no-lean
run --steps 200 --warmup 20 --am ncc --no-lean
run --steps 200 --warmup 20 --am ncc --mode lean --no-lean
run --steps 200 --warmup 20 --targets 1 --no-lean
run --steps 200 --warmup 20 --targets 1 --mode lean --no-lean
run --steps 200 --warmup 20 --targets 1 --sm fclk --mode lean --no-lean
run --steps 200 --warmup 20 --targets 1 --res 50 --mode lean --no-lean
run --steps 200 --warmup 20 --res 50 --mode lean --no-lean
run --steps 50 --warmup 5 --channels 3 --no-lean
run --steps 50 --warmup 5 --channels 3 --am ncc --no-lean
for sm in esm fclk iclk; do timeout 300 python bench.py --workload dropin --sm $sm --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $out; done
timeout 300 python bench.py --workload dropin --sm esm --am ncc --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $out
timeout 300 python bench.py --workload dropin --sm esm --lm 0 --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $out
python - <<'PY'
import json
for l in open("gpurun_out/r02_config_table.jsonl"):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:80]); continue
    r = d.get("roofline") or {}
    print("%-110s %12.0f %8.2f us  k=%s frac=%s" % ((d.get("config") or {}).get("workload", d["metric"])[:110], d["value"], d["ms_per_step"] * 1e3,
          r.get("avg_kernel_ms"), r.get("frac")))
PY
