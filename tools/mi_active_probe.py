import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, mtf_amd
from mtf_amd import synth
from mtf_amd.sm import LKTracker
ctx = mtf_amd.Context(0)
rng = np.random.default_rng(synth.DEFAULT_SEED + 2)
H = W = 2048
frame0 = synth.make_frame(H, W)
p_true = synth.random_small_homography(rng, 0.3)
frame1 = synth.warp_frame(frame0, p_true, (W / 2.0, H / 2.0))
for B in (60, 64, 68):
    res = 400
    rng2 = np.random.default_rng(synth.DEFAULT_SEED + 2); rng2.uniform(size=8)
    half = res / 2.0 + 12
    cx = rng.uniform(half, W - half, size=B); cy = rng.uniform(half, H - half, size=B)
    corners = np.stack([synth.square_corners(cx[i], cy[i], float(res)) for i in range(B)])
    ctx.set_image(frame0)
    nt = LKTracker(ctx, mtf_amd.SM_ESM, mtf_amd.SSM_HOMOGRAPHY, res, res, B, host_solve=False, am=mtf_amd.AM_MI, max_iters=10, epsilon=-1.0, leven_marq=0, materialize=0)
    nt.initialize(corners)
    ctx.set_image(frame1)
    for k in range(4):
        c = nt.update()
        print(B, k, "n_iters", np.bincount(np.asarray(nt.n_iters), minlength=11), "finite", np.isfinite(c).all(axis=(1, 2)).sum())
    nt.batch.close()
