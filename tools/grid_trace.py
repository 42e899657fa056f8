"""Prints the phase durations of one workgroup of k_iclk_track (grid frame, 256 patches x 10 iterations) from a
-DMTFHIP_GRID_TRACE build (tools/grid_trace.sh): MTFHIP_LIB=build/variants/libmtfhip_gtrace.so python tools/grid_trace.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mtf_amd
from mtf_amd import synth, _lib as L
from mtf_amd.sm import GridTracker
ctx = mtf_amd.Context(0)
rng = np.random.default_rng(0)
frame0 = synth.make_frame(1024, 1024)
frame1 = synth.warp_frame(frame0, synth.random_small_homography(rng, 0.3), (512.0, 512.0))
ctx.set_image(frame0)
gt = GridTracker(ctx, grid_size=16, patch_size=25, max_iters=10, epsilon=-1.0)
region = synth.square_corners(512.0, 512.0, 600.0)
gt.initialize(region)
ctx.set_image(frame1)
pc = gt.patch_corners(region)
acc = np.zeros(19)
lib = L.lib()
lib.mtfhip_debug_grid_trace.argtypes = [C.c_void_p]
n = 0
for k in range(60):
    gt.tracker.batch.grid_frame(gt.gd, gt.tracker.sm, region)   # (mtfhip_grid_frame: the kernel lays its patches out itself; MTFHIP_GRID_LAYOUT_DEV=0: the host does)
    if k >= 10:
        t = np.zeros(32, dtype=np.uint64)
        lib.mtfhip_debug_grid_trace(t.ctypes.data_as(C.c_void_p))
        acc += (t[:19].astype(np.float64) - float(t[0])) / 100.0   # 100 MHz -> us
        n += 1
acc /= n
names = ["entry", "prologue loads issued", "first barrier", "tables ready"] + ["iteration %d start" % i for i in range(12)] + ["loop done", "before publish", "after publish"]
prev = 0.0
for i, nm in enumerate(names):
    if i >= 4 + 10 and i < 16: continue
    print("%-24s %7.2f us  (+%.2f)" % (nm, acc[i], acc[i] - prev)); prev = acc[i]
ctx.close()
