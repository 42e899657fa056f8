#!/bin/bash
# NN dataset quick loop on the GPU box: the nn tests, then the bench lines over AM x sample count (kernel time by events, roofline fraction)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out/nn_quick; mkdir -p $o
python -m pytest tests/test_gpu_nn.py tests/test_gpu_golden3.py tests/test_gpu_parity.py -k "nn" -m gpu -x -q 2>&1 | tail -2
: > $o/nn_lines.jsonl
for am in ssd ncc mi; do for n in 1000 10000 100000; do
  [ $am = mi ] && [ $n = 100000 ] && continue
  timeout 200 python bench.py --workload nn --nn-am $am --samples $n --steps 20 --warmup 3 --no-cpu 2>$o/nn_err.txt | tail -1 >> $o/nn_lines.jsonl
done; done
python - <<P
import json
for ln in open("$o/nn_lines.jsonl"):
    d = json.loads(ln); print(d["metric"][27:31], d["config"]["samples_per_rank"], "wall us %.1f" % (d["ms_per_step"] * 1e3), "kernel us %.1f" % (d["roofline"]["avg_kernel_ms"] * 1e3), "frac %.3f" % d["roofline"]["frac"])
P
