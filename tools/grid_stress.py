import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mtf_amd
from mtf_amd import synth
from mtf_amd.sm import GridTracker
ctx = mtf_amd.Context(0)
rng = np.random.default_rng(0)
f0 = synth.make_frame(1024, 1024); f1 = synth.warp_frame(f0, synth.random_small_homography(rng, 0.3), (512.0, 512.0))
ctx.set_image(f0)
gt = GridTracker(ctx, grid_size=16, patch_size=25, max_iters=10, epsilon=-1.0)
region = synth.square_corners(512.0, 512.0, 400.0)
gt.initialize(region); ctx.set_image(f1)
pcs = [gt.patch_corners(region + np.array([[dx], [dy]])) for dx, dy in ((0, 0), (1.25, -0.5), (-2.0, 0.75))]
ref = []
for pc in pcs:
    c, cen = gt.update_patches(pc); ref.append((c.copy(), cen.copy(), gt.n_iters.copy()))
bad = 0
for k in range(6000):
    i = k % 3
    c, cen = gt.update_patches(pcs[i])
    if not (np.array_equal(c, ref[i][0]) and np.array_equal(cen, ref[i][1]) and np.array_equal(gt.n_iters, ref[i][2])):
        bad += 1
print("frames 6000 mismatches", bad)
