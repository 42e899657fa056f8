"""Phase durations of workgroup 0 of k_pf_score from a -DMTFHIP_PF_TRACE build (tools/pf_score_trace.sh)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mtf_amd
from mtf_amd import synth, _lib as L
from mtf_amd.sm import ParticleFilter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ctx = mtf_amd.Context(0)
ctx.set_image(synth.make_frame(1024, 1024))
pf = ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 50, 50, n_particles=n, ssm_sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, seed=5)
pf.initialize(synth.square_corners(512, 512, 100)[None])
lib = L.lib()
lib.mtfhip_debug_pf_trace.argtypes = [C.c_void_p]
acc, m = np.zeros(5), 0
for k in range(60):
    pf.iteration()
    if k >= 10:
        t = np.zeros(16, dtype=np.uint64)
        lib.mtfhip_debug_pf_trace(t.ctypes.data_as(C.c_void_p))
        acc += (t[:5].astype(np.float64) - float(t[0])) / 100.0; m += 1
acc /= m
prev = 0.0
for nm, v in zip(["entry", "warps + hull check done", "pixel loop done", "reduction done", "weights stored"], acc):
    print("%-26s %7.2f us (+%.2f)" % (nm, v, v - prev)); prev = v
ctx.close()
