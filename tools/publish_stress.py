"""the host-publish paths after the fences went: interface-mode iterations (k_finish_host), device-loop track calls (k_publish_host)
and PF iterations (k_pf_select's estimate), thousands of times, each result compared with the first one of its kind"""
import numpy as np, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mtf_amd
from mtf_amd import synth, _lib as L
from mtf_amd.sm import LKTracker, ParticleFilter
ctx = mtf_amd.Context(0)
f0 = synth.make_frame(1024, 1024); f1 = synth.warp_frame(f0, synth.random_small_homography(np.random.default_rng(0), 0.3), (512.0, 512.0))
corners = np.stack([synth.square_corners(300 + 9 * i, 320 + 7 * i, 100) for i in range(16)])
bad = {}
for host_solve in (True, False):
    ctx.set_image(f0)
    sm = LKTracker(ctx, L.SM_ESM, L.SSM_HOMOGRAPHY, 50, 50, 16, host_solve=host_solve, max_iters=4, epsilon=-1.0)
    sm.initialize(corners)
    ctx.set_image(f1)
    ref = None; n_bad = 0
    for k in range(int(os.environ.get("PUBLISH_STRESS_N", "1500"))):
        sm.set_region(corners)
        c = np.array(sm.update())
        if ref is None: ref = c.copy()
        elif not np.array_equal(c, ref): n_bad += 1
    bad["host_solve=%s" % host_solve] = n_bad
ctx.set_image(f0)
pf = ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 50, 50, n_particles=2000, ssm_sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, seed=7)
ref = []
for rep in range(2):
    pf.initialize(corners[:1]); ctx.set_image(f1)
    n_bad = 0
    for k in range(int(os.environ.get("PUBLISH_STRESS_N", "1500"))):
        pf.iteration()
        c = np.array(pf.get_region())
        if rep == 0: ref.append(c.copy())
        elif not np.array_equal(c, ref[k]): n_bad += 1
    ctx.set_image(f0)
bad["pf"] = n_bad
# the grid kernel's own publish (publish_target): one launch per frame
from mtf_amd.sm import GridTracker
ctx.set_image(f0)
gt = GridTracker(ctx, grid_size=16, patch_size=25, max_iters=10, epsilon=-1.0)
region = synth.square_corners(512, 512, 400)
gt.initialize(region); ctx.set_image(f1)
gref = None; n_bad = 0
for k in range(int(os.environ.get("PUBLISH_STRESS_N", "1500"))):
    c, cen = gt.update_patches(region)
    if gref is None: gref = (c.copy(), cen.copy())
    elif not (np.array_equal(c, gref[0]) and np.array_equal(cen, gref[1])): n_bad += 1
bad["grid"] = n_bad
ref = [np.asarray(ref).ravel(), gref[0].ravel()]
ref = np.concatenate(ref)
print("mismatches", bad, "fenced" if os.environ.get("MTFHIP_PUBLISH_FENCE") == "1" else "acknowledged stores")
if os.environ.get("PUBLISH_STRESS_DUMP"):
    np.savez(os.environ["PUBLISH_STRESS_DUMP"], ref=np.asarray(ref), bad=np.asarray([bad[k] for k in sorted(bad)]))
sys.exit(1 if any(bad.values()) else 0)
