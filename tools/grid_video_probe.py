#!/usr/bin/env python
"""microseconds per frame of mtf::hip::Grid::update() in a video loop (setImage + update per frame) in the reset / forward-backward modes.
usage: python tools/grid_video_probe.py [n_frames [substring of the mode names to run]]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mtf_amd
from mtf_amd import host, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
only = sys.argv[2] if len(sys.argv) > 2 else ""
frame0 = synth.make_frame(1024, 1024)
frame1 = synth.warp_frame(frame0, synth.random_small_homography(np.random.default_rng(synth.DEFAULT_SEED + 2), 0.3), (512.0, 512.0))
region = synth.square_corners(512, 512, 400)
for name, kw in (("reset2", dict(reset_at_each_frame=2)), ("reset1", dict(reset_at_each_frame=1)), ("reset0", dict(reset_at_each_frame=0)),
                 ("shipped reset1 fb2 reinit1", dict(reset_at_each_frame=1, fb_err_thresh=2.0, fb_reinit=1)),
                 ("reset1 fb2 reinit0", dict(reset_at_each_frame=1, fb_err_thresh=2.0, fb_reinit=0)),
                 ("reset0 fb2 reinit1", dict(reset_at_each_frame=0, fb_err_thresh=2.0, fb_reinit=1)),
                 ("reset0 fb2 reinit0", dict(reset_at_each_frame=0, fb_err_thresh=2.0, fb_reinit=0))):
    if only and only not in name:
        continue
    g = host.CppGridTracker(grid_size=16, patch_size=25, patch_sm=mtf_amd.SM_ICLK, patch_am=mtf_amd.AM_NCC, patch_ssm=mtf_amd.SSM_AFFINE, grid_ssm=mtf_amd.SSM_HOMOGRAPHY,
                            max_iters=10, epsilon=-1.0, hess_type=0, **kw)
    g.set_image(frame0); g.initialize(region)
    r = [g.bench_video(frame0, frame1, n) for _ in range(3)]
    print("%-28s update %s us  set_image %s us" % (name, " ".join("%.1f" % a for a, _ in r), " ".join("%.1f" % b for _, b in r)), flush=True)
    del g
