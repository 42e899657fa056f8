#!/bin/bash
# Collect PMC counters for the bench workload, one rocprofv3 --pmc pass per counter group
# (no trace domains combined with --pmc).  Usage: tools/pmc_collect.sh <outdir> "<bench args>" "<group1>" "<group2>" ...
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=$1; shift
args=$1; shift
mkdir -p "$out"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -o pmc -- python bench.py $args --no-cpu > "$out/p$i.log" 2>&1
done
python tools/pmc_summary.py "$out" > "$out/summary.txt"
cat "$out/summary.txt"
