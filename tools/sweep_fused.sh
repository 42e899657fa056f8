#!/bin/bash
# Build fused-kernel variants (MTFHIP_FUSED_WAVES / MTFHIP_MIN_ROWS / MTFHIP_SLOTS; extra -D flags through SWEEP_EXTRA) and bench each on the GPU box.
# usage: tools/sweep_fused.sh "<pipe> <waves> <min_rows> [coop] [slots]" ...   (A/B only within ONE call: boxes differ by 10-15 %)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sweep
for v in "$@"; do
  set -- $v
  pipe=$1; waves=$2; ppt=$3; coop=${4:-0}; slots=${5:-512}
  so=/tmp/libmtfhip_${pipe}_${waves}_${ppt}_${coop}_${slots}.so
  make -C mtf_amd/csrc -s -B -j8 OUT=$so EXTRA="-DMTFHIP_FUSED_WAVES=$waves -DMTFHIP_MIN_ROWS=$ppt -DMTFHIP_SLOTS=$slots $SWEEP_EXTRA" 2>&1 | grep -E "error" | head -3
  for mode in full lean; do
    MTFHIP_LIB=$so python bench.py --steps 40 --warmup 8 --no-cpu --mode $mode 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('pipe=$pipe waves=$waves min_rows=$ppt coop=$coop slots=$slots', d['config']['mode'], 'iters/s', round(d['value']), 'kern_ms %.4f' % d['roofline']['avg_kernel_ms'], 'frac %.3f' % d['roofline']['frac'])"
  done
done
