#!/bin/bash
# Ablation of k_mi_pass_hist (pass 1 of the MI recompute iteration): variants of libmtfhip.so with -DMTFHIP_MI1_ABL=n under build/variants/
# (built HERE on the CPU; build/ travels to the GPU box, scratch/ does not).  On the box:  bash tools/mi1_ablation.sh run
#   1 no histogram product | 2 staging but no block products | 3 windows but no staging | 4 sampling only
if [ "$1" = run ]; then
  cd "$GRAFT_REPO_ROOT" || exit 1
  for n in 0 1 2 3 4; do
    lib=mtf_amd/libmtfhip.so; [ $n != 0 ] && lib=build/variants/libmtfhip_mi1abl$n.so
    echo -n "MI1_ABL=$n: "
    MTFHIP_LIB=$lib python bench.py --workload mi --steps 5 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']['avg_kernel_ms']
print('pass1 %.1f us  pass2 %.1f us' % (r['pass1'] * 1e3, r['pass2'] * 1e3))"
  done
  exit 0
fi
cd "$(dirname "$0")/../mtf_amd/csrc" || exit 1
mkdir -p ../../build/variants
for n in 1 2 3 4; do
  make -s -j8 OUT=../../build/variants/libmtfhip_mi1abl$n.so EXTRA="-DMTFHIP_MI1_ABL=$n" || exit 1
done
