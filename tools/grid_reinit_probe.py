"""where a reset_at_each_frame = 1 frame goes: HIP-event durations of the launches of one frame + host wall times of the two halves"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import mtf_amd
from mtf_amd import synth
from mtf_amd.sm import GridTracker
ctx = mtf_amd.Context(0)
frame0 = synth.make_frame(1024, 1024)
frame1 = synth.warp_frame(frame0, synth.random_small_homography(np.random.default_rng(1), 0.3), (512.0, 512.0))
region = synth.square_corners(512, 512, 400)
ctx.set_image(frame0)
g = GridTracker(ctx, grid_size=16, patch_size=25, max_iters=10, epsilon=-1.0, reset_at_each_frame=1)
g.initialize(region); ctx.set_image(frame1)
b, sm, gd = g.tracker.batch, g.tracker.sm, g.gd
for k in range(30):
    b.grid_frame(gd, sm, None); b.grid_reset(gd, sm, region, True)
torch.cuda.synchronize()
n = 300
t_f = t_r = 0.0
for k in range(n):
    t0 = time.perf_counter(); b.grid_frame(gd, sm, None); t1 = time.perf_counter(); b.grid_reset(gd, sm, region, True); t2 = time.perf_counter()
    t_f += t1 - t0; t_r += t2 - t1
torch.cuda.synchronize()
print("host wall per frame: grid_frame (track, waits for the corners) %.1f us, grid_reset(reinit) call %.1f us (returns without waiting)" % (t_f / n * 1e6, t_r / n * 1e6))
ctx.timing(True); ctx.timing_reset()
for k in range(100):
    b.grid_frame(gd, sm, None); b.grid_reset(gd, sm, region, True)
torch.cuda.synchronize(); ctx.timing(False)
for fam in ("iclk_track", "template_init", "init_grid", "apply_warp"):
    ms, cnt = ctx.timing_get(fam)
    print("kernel family %-14s avg %.1f us over %d launches" % (fam, ms * 1e3, cnt))
