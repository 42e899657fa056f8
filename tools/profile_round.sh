#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench command, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE and a small SQ/TCC set) of the same command.  Outputs go to gpurun_out/profile/
# (copy the summaries into profiles/ afterwards).   usage: tools/profile_round.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
args="${*:---steps 50 --warmup 10}"
out=gpurun_out/profile
mkdir -p $out
# (the PMC passes come first: the bench line quotes their traffic figure, and only for the kernel sources it was taken on)
python bench.py $args --no-cpu --no-lean > $out/${tag}_bench_pre.json 2> $out/${tag}_bench.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  # device-wide counters per dispatch: the two-queue loop is serialised for these passes (same launches, one queue)
  MTFHIP_TRACK_SERIALIZE=1 timeout 400 rocprofv3 --pmc $grp --output-format csv -d $out/pmc/p$i -o pmc -- python bench.py --steps 10 --warmup 2 --no-cpu --no-lean ${args##*--warmup [0-9]*} > $out/pmc_p$i.log 2>&1
done
python tools/pmc_summary.py $out/pmc > $out/${tag}_pmc_summary.txt
python - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(list)
for f in glob.glob("$out/pmc/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fused_ssd" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
f = sum(agg["FETCH_SIZE"]) / max(1, len(agg["FETCH_SIZE"]))
w = sum(agg["WRITE_SIZE"]) / max(1, len(agg["WRITE_SIZE"]))
# MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
import os, sys, subprocess
sys.path.insert(0, ".")
import bench
commit = open(".git_head").read().strip() if os.path.exists(".git_head") else None
try:
    per_launch = json.loads(open("$out/${tag}_bench_pre.json").read().strip().splitlines()[-1])["roofline"]["targets_per_launch"]
except Exception:
    per_launch = None
json.dump({"kernel": "k_fused_ssd", "bench_args": "$args", "commit": commit, "targets_per_launch": per_launch, "kernel_sources_sha": bench.kernel_sources_sha(), "j0_recompute": os.environ.get("MTFHIP_J0_RECOMPUTE", "1") != "0", "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
           "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count, MI355X_MICROARCH.md HBM section)"},
          open("$out/${tag}_pmc_traffic.json", "w"), indent=1)
PY
cp $out/${tag}_pmc_traffic.json profiles/pmc_latest.json
python bench.py $args > $out/${tag}_bench.json 2>> $out/${tag}_bench.err
# (--no-lean: the lean and single-target sub-records launch the SAME kernels at other sizes and would be averaged into the headline's rows)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- python bench.py $args --no-cpu --no-lean > $out/${tag}_trace.log 2>&1
cp $out/trace/${tag}_kernel_stats.csv $out/${tag}_kernel_stats.csv
# the same command with the loop on one queue: the kernel with the device to itself (r01 / r02 figures are of this form)
MTFHIP_TRACK_STREAMS=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1q -o ${tag}_1q -- python bench.py $args --no-cpu --no-lean > $out/${tag}_trace1q.log 2>&1
cp $out/trace1q/${tag}_1q_kernel_stats.csv $out/${tag}_kernel_stats_one_queue.csv
MTFHIP_TRACK_STREAMS=1 python bench.py $args --no-cpu --no-lean > $out/${tag}_bench_one_queue.json 2>> $out/${tag}_bench.err
cat $out/${tag}_bench.json | cut -c1-400; head -5 $out/${tag}_kernel_stats.csv | cut -c1-200; cat $out/${tag}_pmc_traffic.json
