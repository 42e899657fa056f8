#!/bin/bash
# r06: ONE evidence run on the final sources (r05 verdict item 8), one gpurun call: gpurun_out/r06/
#   1 headline: bench line, rocprofv3 kernel trace, PMC traffic passes            2 PMC passes of the secondary workloads (mi, pf, grid, nn)
#   3 the driver's own command (the line with the configs block)                   4 secondary bench lines + kernel traces (grid incl. the shipped
#   forward-backward frame, pf, mi 8 / 10 bins, nn over AM x size)                  5 the parity record of the -m gpu tests
# Every rocprofv3 call runs under `timeout` (a hung counter pass cost 15 GPU-minutes once).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
tag=r06
out=gpurun_out/$tag
mkdir -p $out
# 1. headline (tools/profile_round.sh writes gpurun_out/profile/ and profiles/pmc_latest.json)
bash tools/profile_round.sh $tag --steps 50 --warmup 10 > $out/profile_round.log 2>&1
cp gpurun_out/profile/${tag}_* $out/ 2>/dev/null
cp profiles/pmc_latest.json $out/pmc_latest.json
# 2. PMC of the secondary workloads' kernels (separate passes, no trace domains) -> the file their bench lines quote
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_mi/p$i -o pmc -- python bench.py --workload mi --steps 3 --warmup 1 --no-cpu > $out/pmc_mi_p$i.log 2>&1
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_pf/p$i -o pmc -- python bench.py --workload pf --steps 10 --warmup 2 --no-cpu > $out/pmc_pf_p$i.log 2>&1
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_grid/p$i -o pmc -- python bench.py --workload grid --steps 20 --warmup 2 --no-cpu > $out/pmc_grid_p$i.log 2>&1
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_nn/p$i -o pmc -- python bench.py --workload nn --steps 5 --warmup 1 --no-cpu > $out/pmc_nn_p$i.log 2>&1
done
for wl in mi pf grid nn; do python tools/pmc_summary.py $out/pmc_$wl > $out/${wl}_pmc_summary.txt; done
python tools/pmc_secondary_json.py $out profiles/pmc_secondary_latest.json && cp profiles/pmc_secondary_latest.json $out/pmc_secondary_latest.json
rm -rf $out/pmc_mi $out/pmc_pf $out/pmc_grid $out/pmc_nn
# 3. the driver's own command (configs block included), twice
: > $out/final_bench_lines.jsonl
for k in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $out/final_bench_lines.jsonl; done
# 4. secondary workloads: full lines + kernel traces
: > $out/secondary_bench_lines.jsonl
trace() { # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -o $name -- python bench.py "$@" --no-cpu > $out/${name}_trace.log 2>&1
  find $out/trace_$name -name '*kernel_stats.csv' -exec cp {} $out/${name}_kernel_stats.csv \;
  rm -rf $out/trace_$name
}
timeout 300 python bench.py --workload grid --steps 200 --warmup 10 --cpu-seconds 3 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
trace grid --workload grid --steps 200 --warmup 10
timeout 300 python bench.py --workload pf --particles 10000 --steps 200 --warmup 10 --cpu-seconds 3 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
timeout 300 python bench.py --workload pf --particles 10000 --pf-iters 10 --steps 50 --warmup 5 --no-cpu 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
trace pf_chained --workload pf --particles 10000 --pf-iters 10 --steps 50 --warmup 5
for n in 100000 1000000; do timeout 300 python bench.py --workload pf --particles $n --steps 30 --warmup 3 --no-cpu 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl; done
timeout 300 python bench.py --workload mi --steps 5 --warmup 2 --cpu-seconds 3 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
trace mi --workload mi --steps 5 --warmup 2
timeout 300 python bench.py --workload mi --mi-bins 10 --mi-pou 1 --steps 5 --warmup 2 --cpu-seconds 3 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
trace mi10 --workload mi --mi-bins 10 --mi-pou 1 --steps 5 --warmup 2
for am in ssd ncc mi; do for n in 1000 10000 100000; do
  [ $am = mi ] && [ $n = 100000 ] && continue
  timeout 200 python bench.py --workload nn --nn-am $am --samples $n --steps 20 --warmup 3 --cpu-seconds 2 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
done; done
trace nn --workload nn --steps 20 --warmup 3
timeout 300 python bench.py --workload dropin --sm esm --steps 200 --warmup 20 --cpu-seconds 3 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
timeout 300 python bench.py --workload dropin --sm esm --device-loop --steps 200 --warmup 20 --no-cpu 2>/dev/null | tail -1 >> $out/secondary_bench_lines.jsonl
timeout 600 python bench.py --pf-strong 1 --steps 30 --warmup 5 --no-cpu --no-lean --configs 0 2>/dev/null | tail -1 > $out/pf_strong_one_rank.json
# the shipped grid frame: one launch against three (same box)
: > $out/grid_fb_ab.txt
for f in 1 0 1 0; do MTFHIP_GRID_FB_FUSED=$f timeout 200 python bench.py --workload grid --steps 100 --warmup 10 --no-cpu 2>/dev/null | tail -1 |
  python -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['config']['cpp_driver']['video_loop']; print('MTFHIP_GRID_FB_FUSED=$f  shipped (reset 1, fb 2, reinit 1) update() %.1f us, set_image %.1f us; reset1 without fb %.1f us' % (v['shipped_reset1_fb2_reinit1']['update_us'], v['shipped_reset1_fb2_reinit1']['set_image_us'], v['reset1_reinit']['update_us']))" >> $out/grid_fb_ab.txt; done
# 5. the parity record of the GPU tests (measured errors, not just pass / fail)
MTFHIP_PARITY_RECORD=$out/parity_record.jsonl timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $out/gpu_tests.log
cat $out/gpu_tests.log
python - <<'PY'
import json
for f in ("final_bench_lines", "secondary_bench_lines"):
    print("==", f)
    for l in open("gpurun_out/r06/%s.jsonl" % f):
        try: d = json.loads(l)
        except Exception: print("bad line", l[:80]); continue
        r = d.get("roofline") or {}
        print("%-90s %14.0f %9.2f us k=%s frac=%s" % ((d.get("config") or {}).get("workload", d["metric"])[:90], d["value"], d["ms_per_step"] * 1e3, r.get("avg_kernel_ms"), r.get("frac")))
        for k, v in (d.get("configs") or {}).items():
            if isinstance(v, dict): print("      %-44s %s %s %s" % (k, v.get("value"), v.get("unit"), v.get("error", "")))
PY
cat $out/grid_fb_ab.txt
head -8 $out/${tag}_kernel_stats.csv | cut -c1-200; cat $out/${tag}_pmc_traffic.json
