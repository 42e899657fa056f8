"""Phase durations of k_pf_select (10 000 particles) from a -DMTFHIP_PF_TRACE build (tools/pf_trace.sh)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mtf_amd
from mtf_amd import synth, _lib as L
from mtf_amd.sm import ParticleFilter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = mtf_amd.Context(0)
frame = synth.make_frame(512, 512)
ctx.set_image(frame)
pf = ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 50, 50, n_particles=n, ssm_sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, seed=5)
pf.initialize(synth.square_corners(256, 256, 100)[None])
lib = L.lib()
lib.mtfhip_debug_pf_trace.argtypes = [C.c_void_p]
acc0, acc1, m = np.zeros(6), np.zeros(4), 0
for k in range(60):
    pf.iteration()
    if k >= 10:
        t = np.zeros(16, dtype=np.uint64)
        lib.mtfhip_debug_pf_trace(t.ctypes.data_as(C.c_void_p))
        t = t.astype(np.float64) / 100.0
        acc0 += t[:6] - t[0]; acc1 += t[8:12] - t[0]; m += 1
acc0 /= m; acc1 /= m
for nm, v in zip(["entry (workgroup 0)", "chunk table in LDS", "search done", "copy + look-ahead proposal done", "partial row stored + acked", "arrived"], acc0):
    print("%-34s %7.2f us" % (nm, v))
for nm, v in zip(["last arriver: fold starts", "fold done", "estimate in device memory", "published to the host"], acc1):
    print("%-34s %7.2f us (since workgroup 0's entry)" % (nm, v))
ctx.close()
