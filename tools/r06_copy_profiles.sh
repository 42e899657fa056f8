#!/bin/bash
# copies the summaries of gpurun_out/r06 (tools/r06_final.sh) into profiles/ under their r06_ names
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r06
for f in r06_bench.json r06_bench_one_queue.json r06_kernel_stats.csv r06_kernel_stats_one_queue.csv r06_pmc_summary.txt r06_pmc_traffic.json; do cp $o/$f profiles/$f; done
for f in final_bench_lines.jsonl secondary_bench_lines.jsonl parity_record.jsonl pf_strong_one_rank.json grid_fb_ab.txt grid_kernel_stats.csv pf_chained_kernel_stats.csv mi_kernel_stats.csv mi10_kernel_stats.csv nn_kernel_stats.csv mi_pmc_summary.txt pf_pmc_summary.txt grid_pmc_summary.txt nn_pmc_summary.txt gpu_tests.log; do cp $o/$f profiles/r06_$f; done
cp $o/pmc_latest.json profiles/pmc_latest.json; cp $o/pmc_secondary_latest.json profiles/pmc_secondary_latest.json
ls profiles | grep r06 | wc -l
