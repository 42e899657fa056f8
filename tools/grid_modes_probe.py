"""frame time of mtf::hip::Grid::update() (C++ loop) in the three reset modes, and of the two resets alone"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import mtf_amd
from mtf_amd import host, synth
frame0 = synth.make_frame(1024, 1024)
frame1 = synth.warp_frame(frame0, synth.random_small_homography(np.random.default_rng(1), 0.3), (512.0, 512.0))
region = synth.square_corners(512, 512, 400)
for mode in (0, 2, 1):
    cg = host.CppGridTracker(grid_size=16, patch_size=25, reset_at_each_frame=mode, max_iters=10, epsilon=-1.0, hess_type=0)
    cg.set_image(frame0); cg.initialize(region); cg.set_image(frame1)
    print("reset_at_each_frame=%d: Grid::update() %.1f us per frame; mtfhip_grid_frame alone %.1f us" % (mode, cg.bench_frames(region, 200, 1), cg.bench_frames(region, 200, 0)))
    del cg
