#!/bin/bash
# r05: everything the round's figures come from, one gpurun call (boxes differ by 10-15 %): gpurun_out/r05/
#   headline bench line + rocprofv3 kernel trace + PMC passes (traffic) | PMC passes of the secondary workloads (mi, pf, grid) ->
#   pmc_secondary_latest.json, which their bench lines then quote | secondary bench lines with kernel traces | the grid / PF probes of
#   the round | the configuration table | the parity record of the -m gpu tests
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r05
out=gpurun_out/$tag
mkdir -p $out
# 1. headline: bench line, kernel trace, PMC traffic (tools/profile_round.sh writes gpurun_out/profile/ and profiles/pmc_latest.json)
bash tools/profile_round.sh $tag --steps 50 --warmup 10 > $out/profile_round.log 2>&1
cp gpurun_out/profile/${tag}_* $out/ 2>/dev/null
cp profiles/pmc_latest.json $out/pmc_latest.json
# 2. PMC of the secondary workloads' kernels (separate passes, no trace domains) -> the file their bench lines quote
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_mi/p$i -o pmc -- python bench.py --workload mi --steps 3 --warmup 1 --no-cpu > $out/pmc_mi_p$i.log 2>&1
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_pf/p$i -o pmc -- python bench.py --workload pf --steps 10 --warmup 2 --no-cpu > $out/pmc_pf_p$i.log 2>&1
  timeout 400 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_grid/p$i -o pmc -- python bench.py --workload grid --steps 20 --warmup 2 --no-cpu > $out/pmc_grid_p$i.log 2>&1
done
python tools/pmc_summary.py $out/pmc_mi > $out/mi_pmc_summary.txt
python tools/pmc_summary.py $out/pmc_pf > $out/pf_pmc_summary.txt
python tools/pmc_summary.py $out/pmc_grid > $out/grid_pmc_summary.txt
python tools/pmc_secondary_json.py $out profiles/pmc_secondary_latest.json && cp profiles/pmc_secondary_latest.json $out/pmc_secondary_latest.json
# the driver's own command, three times (box-internal variance)
: > $out/final_bench_lines.jsonl
timeout 600 python bench.py 2>/dev/null | tail -1 >> $out/final_bench_lines.jsonl
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> $out/final_bench_lines.jsonl; done
# 3. secondary workloads
: > $out/secondary_bench_lines.jsonl
for wl in grid pf mi; do
  steps=200; [ $wl = mi ] && steps=5
  timeout 600 python bench.py --workload $wl --steps $steps --warmup 5 --cpu-seconds 4 2>/dev/null | tail -1 | tee $out/${wl}_bench.json >> $out/secondary_bench_lines.jsonl
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$wl -o $wl -- python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu > $out/${wl}_trace.log 2>&1
  find $out/trace_$wl -name '*kernel_stats.csv' -exec cp {} $out/${wl}_kernel_stats.csv \;
done
for n in 100000 1000000; do timeout 600 python bench.py --workload pf --particles $n --steps 50 --warmup 5 --no-cpu 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl; done
timeout 600 python bench.py --workload pf --particles 10000 --pf-iters 10 --steps 50 --warmup 5 --no-cpu 2>/dev/null | tail -1 | tee $out/pf_chained_bench.json >> $out/secondary_bench_lines.jsonl
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_pfc -o pfc -- python bench.py --workload pf --particles 10000 --pf-iters 10 --steps 50 --warmup 5 --no-cpu > $out/pfc_trace.log 2>&1
find $out/trace_pfc -name '*kernel_stats.csv' -exec cp {} $out/pf_chained_kernel_stats.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_pf1m -o pf1m -- python bench.py --workload pf --particles 1000000 --steps 20 --warmup 3 --no-cpu > $out/pf1m_trace.log 2>&1
find $out/trace_pf1m -name '*kernel_stats.csv' -exec cp {} $out/pf1m_kernel_stats.csv \;
timeout 300 python bench.py --workload dropin --sm esm --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
timeout 300 python bench.py --workload dropin --sm esm --device-loop --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
# MI at the other target counts the round quotes (explicit --res: the default resolution depends on --targets)
for t in 16 32 64; do timeout 300 python bench.py --workload mi --targets $t --res 400 --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl; done
# 3b. the round's A/Bs and probes: PF selection forms, the grid's three reset modes and the fused re-initialisation, the sharded filter on one rank
: > $out/pf_forms_ab.jsonl
for cfg in "1 1 1" "0 1 1" "1 0 1" "1 1 0" "0 0 0"; do
  set -- $cfg
  MTFHIP_PF_LOCAL=$1 MTFHIP_PF_PERT_AHEAD=$2 MTFHIP_PF_SKIP_ESTIMATE=$3 timeout 300 python bench.py --workload pf --particles 10000 --pf-iters 10 --steps 50 --warmup 5 --no-cpu 2>/dev/null | tail -1 |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(json.dumps({'local': $1, 'pert_ahead': $2, 'skip_estimate': $3, 'us_per_iteration': c['us_per_iteration'], 'score_us': c['score_kernel_ms']*1e3, 'resample_us': c['resample_kernels_ms']*1e3}))" >> $out/pf_forms_ab.jsonl
done
timeout 300 python tools/grid_modes_probe.py 2>/dev/null > $out/grid_modes_probe.txt
# the grid frame with the host layout in front of the launch / the kernel's own layout (C++ loop and Python), and the frame's host-side stamps
: > $out/grid_layout_ab.txt
for d in 1 0 1 0; do MTFHIP_GRID_LAYOUT_DEV=$d timeout 300 python bench.py --workload grid --steps 300 --warmup 20 --no-cpu 2>/dev/null | tail -1 |
  python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('MTFHIP_GRID_LAYOUT_DEV=$d frame %.2f us kernel %.2f us; C++ loop %.2f us, Grid::update() %.2f us' % ((c.get('python_wrapper_loop') or c)['frame_us'], c['kernel_us'], c['cpp_driver']['frame_us_c_abi_loop'], c['cpp_driver']['frame_us_grid_update_setregion_mode']))" >> $out/grid_layout_ab.txt; done
MTFHIP_TRACK_DEBUG_TIMING=1 timeout 300 python tools/grid_modes_probe.py 2>&1 | grep track_region | tail -3 >> $out/grid_layout_ab.txt
# MI with partition of unity (the shipped mi_pou = 1): histogram as the joint histogram's row sums / as its own block product
: > $out/mi_pou_ab.txt
for r in 1 0 1 0; do MTFHIP_MI_HIST_ROWSUM=$r timeout 300 python tools/mi_pou_probe.py 2>/dev/null | tail -1 >> $out/mi_pou_ab.txt; done
timeout 300 python tools/grid_reinit_probe.py 2>/dev/null > $out/grid_reinit_probe.txt
timeout 600 python bench.py --pf-strong 1 --steps 50 --warmup 10 --no-cpu --no-lean 2>/dev/null | tail -1 > $out/pf_strong_one_rank.json
timeout 300 python tools/pf_peer_probe.py 2>/dev/null > $out/pf_peer_probe.json
# 4. configuration table
ct=$out/config_table.jsonl; : > $ct
run() { timeout 300 python bench.py "$@" --no-cpu 2>/dev/null | tail -1 >> $ct; }
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
for sm in esm fclk iclk; do timeout 300 python bench.py --workload dropin --sm $sm --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $ct; done
timeout 300 python bench.py --workload dropin --sm esm --am ncc --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $ct
# 5. the parity record of the GPU tests (measured errors, not just pass / fail)
MTFHIP_PARITY_RECORD=$out/parity_record.jsonl timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $out/gpu_tests.log
cat $out/gpu_tests.log
python - <<'PY'
import json, glob
for f in ("final_bench_lines", "secondary_bench_lines", "config_table"):
    print("==", f)
    for l in open("gpurun_out/r05/%s.jsonl" % f):
        try: d = json.loads(l)
        except Exception: print("bad line", l[:80]); continue
        r = d.get("roofline") or {}
        print("%-100s %12.0f %9.2f us k=%s frac=%s lean=%s" % ((d.get("config") or {}).get("workload", d["metric"])[:100], d["value"], d["ms_per_step"] * 1e3,
              r.get("avg_kernel_ms"), r.get("frac"), (d.get("lean") or {}).get("value")))
PY
cat $out/pf_forms_ab.jsonl; cat $out/grid_modes_probe.txt; cat $out/grid_layout_ab.txt; cat $out/mi_pou_ab.txt
head -6 $out/r05_kernel_stats.csv | cut -c1-200; cat $out/r05_pmc_traffic.json
