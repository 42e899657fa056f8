#!/bin/bash
# r03 A/B: LDS slab row strides of the MI recompute passes (bank conflicts of the per-lane-row window stores vs the MFMA operand reads)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in ${MI_AB_VARIANTS:-"68 100" "65 100" "66 100"}; do
  set -- ${v//:/ }
  so=/tmp/libmtfhip_mi_$1_$2_${3:-64}.so
  make -C mtf_amd/csrc -s -B -j8 OUT=$so EXTRA="-DMTFHIP_MI_RS=$1 -DMTFHIP_MI_RS2=$2 -DMTFHIP_MI_QR=${3:-64}" 2>&1 | grep -E "error" | head -3
  MTFHIP_LIB=$so python bench.py --workload mi --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']['avg_kernel_ms']
print('RS=$1 RS2=$2 QR=${3:-64} target-iters/s', round(d['value']), 'pass1 %.1f us pass2 %.1f us' % (r['pass1']*1e3, r['pass2']*1e3))" | tee -a gpurun_out/r03_mi_stride_ab.txt
done
