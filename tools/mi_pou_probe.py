import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, mtf_amd
from mtf_amd import synth, _lib as L
ctx = mtf_amd.Context(0)
rng = np.random.default_rng(3)
f0 = synth.make_frame(1024, 1024); ctx.set_image(f0)
B, res = 64, 400
b = mtf_amd.Batch(ctx, L.AM_MI, L.SSM_HOMOGRAPHY, res, res, B, mi_pou=1)
cs = np.stack([synth.square_corners(512 + rng.uniform(-20, 20), 512 + rng.uniform(-20, 20), 400.0) for _ in range(B)])
b.set_corners(cs)
sm = mtf_amd.sm_desc(L.SM_ESM, leven_marq=0, max_iters=10, epsilon=-1.0, materialize=0)
b.init_template(sm)
b.set_math_mode(mtf_amd.MATH_FAST)
ctx.set_image(synth.warp_frame(f0, synth.random_small_homography(rng, 0.3), (512.0, 512.0)))
for _ in range(3): b.set_corners(cs); b.track(sm)
ctx.timing(True); ctx.timing_reset()
t0 = time.perf_counter()
for _ in range(5): b.set_corners(cs); b.track(sm)
dt = time.perf_counter() - t0
ctx.timing(False)
print("rowsum=%s  %.0f target-iters/s; families: %s" % (os.environ.get("MTFHIP_MI_HIST_ROWSUM", "1"), B * 10 * 5 / dt,
      {k: round(ctx.timing_get(k)[0] * 1e3, 1) for k in ("mi_pass1", "mi_pass2", "mi_hist", "mi_pass_hist")}))
