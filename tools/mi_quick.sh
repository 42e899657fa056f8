python -m pytest tests/test_gpu_parity.py tests/test_gpu_trackers.py tests/test_gpu_golden.py tests/test_gpu_golden2.py tests/test_gpu_fullsize.py -m gpu -q -x -k "mi or MI" 2>&1 | tail -4
python bench.py --workload mi --steps 5 --warmup 5 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('MI value %.0f  ms/step %.3f  pass1 %.1f us  pass2 %.1f us' % (d['value'], d['ms_per_step'], r['avg_kernel_ms']['pass1'] * 1e3, r['avg_kernel_ms']['pass2'] * 1e3))"
