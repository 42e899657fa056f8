#!/bin/bash
# times the MI passes with every build/variants/libmtfhip_*.so (and the in-tree library first)
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in mtf_amd/libmtfhip.so build/variants/libmtfhip_*.so; do
  echo -n "$(basename $lib): "
  MTFHIP_LIB=$lib python bench.py --workload mi --steps 5 --warmup 5 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']['avg_kernel_ms']
print('pass1 %.1f us  pass2 %.1f us  value %.0f' % (r['pass1'] * 1e3, r['pass2'] * 1e3, d['value']))"
done
