#!/bin/bash
# Kernel traces + PMC passes of the secondary workloads (grid, pf, mi) on the GPU box.  Outputs: gpurun_out/<tag>/
# usage: tools/r02_secondary_profiles.sh <tag>     (copy the summaries into profiles/ afterwards)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
for wl in grid pf mi; do
  steps=100; [ $wl = mi ] && steps=5
  python bench.py --workload $wl --steps $steps --warmup 5 --cpu-seconds 4 > $out/${wl}_bench.json 2> $out/${wl}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$wl -o $wl -- python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu > $out/${wl}_trace.log 2>&1
  cp $out/trace_$wl/${wl}_kernel_stats.csv $out/${wl}_kernel_stats.csv 2>/dev/null || find $out/trace_$wl -name '*kernel_stats.csv' -exec cp {} $out/${wl}_kernel_stats.csv \;
done
# PMC of the candidate scorer (separate passes, no trace domains)
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_pf/p$i -o pmc -- python bench.py --workload pf --steps 10 --warmup 2 --no-cpu > $out/pmc_pf_p$i.log 2>&1
done
python tools/pmc_summary.py $out/pmc_pf > $out/pf_pmc_summary.txt
head -12 $out/*_kernel_stats.csv | cut -c1-180
cat $out/pf_pmc_summary.txt
cat $out/*_bench.json | cut -c1-600
