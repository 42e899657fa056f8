import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, mtf_amd
from mtf_amd import synth, _lib as L
ctx = mtf_amd.Context(0)
f0 = synth.make_frame(1024, 1024); ctx.set_image(f0)
for smk, name in ((L.SM_ESM, "esm"), (L.SM_FCLK, "fclk"), (L.SM_ICLK, "iclk")):
    b = mtf_amd.Batch(ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, 1)
    c = synth.square_corners(512, 512, 200.0)[None]
    b.set_corners(c)
    sm = mtf_amd.sm_desc(smk, leven_marq=int(os.environ.get("LM", "1")), max_iters=50, epsilon=-1.0, materialize=0)
    b.init_template(sm)
    for _ in range(5): b.set_region(c + 0.1, sm); b.track(sm)
    t0 = time.perf_counter()
    for k in range(50): b.set_region(c + 0.01 * k, sm)
    t1 = time.perf_counter()
    for k in range(50): b.track(sm)
    t2 = time.perf_counter()
    for k in range(50): b.set_region(c + 0.01 * k, sm); b.track(sm)
    t3 = time.perf_counter()
    print("%s: set_region %.1f us, track(50 passes) %.1f us, both %.1f us" % (name, (t1 - t0) / 50 * 1e6, (t2 - t1) / 50 * 1e6, (t3 - t2) / 50 * 1e6))
    b.close()
