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
# r05: the same through mtfhip_grid_frame (the kernel lays its own patch out from the grid's region, the host lays the patches out behind the
# launch) with a fused re-initialisation (mtfhip_grid_reset, reinit) every 50 frames: every frame and every re-initialised template against
# the first occurrence of its (region, frame-since-reinit) pair
b, gd, sm = gt.tracker.batch, gt.gd, gt.tracker.sm
regions = [region + np.array([[dx], [dy]]) for dx, dy in ((0, 0), (1.25, -0.5), (-2.0, 0.75))]
ref2, bad2, n2 = {}, 0, 0
ctx.set_image(f0); b.grid_reset(gd, sm, regions[0], True); ctx.set_image(f1)
for k in range(3000):
    i = k % 3
    if k % 50 == 0:
        ctx.set_image(f0); pcs_r, pp = b.grid_reset(gd, sm, regions[i], True); ctx.set_image(f1)
        key, val = ("reinit", i), (pcs_r.copy(), pp.copy(), b.read(mtf_amd._lib.BUF_I0).copy())
    else:
        n, c, m = b.grid_frame(gd, sm, regions[i])
        key, val = ("frame", i, (k // 50) % 3), (n.copy(), c.copy(), m.copy())
    if key in ref2:
        n2 += 1
        if not all(np.array_equal(x, y) for x, y in zip(val, ref2[key])): bad2 += 1
    else:
        ref2[key] = val
print("grid_frame / grid_reset(reinit) frames compared", n2, "mismatches", bad2)
