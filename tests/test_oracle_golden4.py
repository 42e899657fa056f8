"""r06: GridTracker's forward-backward error estimation (SM/src/GridTracker.cc:186-190, 241-243, 263-266, 294-343) on the CPU side:
the selection half (mask + the point pairs handed to the estimator) of the C++ oracle AND of the C ABI's host function
mtfhip_grid_fb_mask against tests/golden/lk_golden4.npz (independent NumPy re-derivation, tests/golden/make_golden4.py), and the
oracle's whole backwardEstimation over its own per-patch trackers through properties the domain offers (a round trip on an unchanged
scene comes back; a patch whose content was replaced between the frames does not)."""
import ctypes as C
import os

import numpy as np
import pytest

from mtf_amd import _lib as L
from mtf_amd import synth
from mtf_amd.sm import least_squares_estimator

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden4.npz"))
CASES = [str(s) for s in G["fb_case_names"]]


@pytest.mark.parametrize("case", CASES)
def test_fb_mask_golden(oracle, case):
    prev, curr, fb = G["fb_%s_prev" % case], G["fb_%s_curr" % case], G["fb_%s_fb" % case]
    thresh, nm = float(G["fb_%s_params" % case][0]), int(G["fb_%s_params" % case][1])
    want_mask, want_p, want_c = G["fb_%s_mask" % case], G["fb_%s_prev_masked" % case], G["fb_%s_curr_masked" % case]
    # the oracle's restatement
    mask, pm, cm = oracle.grid_fb_mask(prev, curr, fb, thresh, nm)
    assert np.array_equal(mask, want_mask)
    assert np.array_equal(pm, want_p) and np.array_equal(cm, want_c)
    # the product's host arithmetic (no device needed)
    lib = C.CDLL(L.LIB_PATH)
    n = len(prev)
    fbd = L.GridFbDesc(thresh, 1, nm)
    m2, p2, c2, cnt = np.zeros(n, dtype=np.uint8), np.zeros((n, 2), dtype=np.float32), np.zeros((n, 2), dtype=np.float32), C.c_int()
    lib.mtfhip_grid_fb_mask.argtypes = [C.c_int] + [C.c_void_p] * 8
    a, b, f = (np.ascontiguousarray(x, dtype=np.float32) for x in (prev, curr, fb))
    assert lib.mtfhip_grid_fb_mask(n, a.ctypes.data, b.ctypes.data, f.ctypes.data, C.addressof(fbd), m2.ctypes.data, p2.ctypes.data, c2.ctypes.data, C.addressof(cnt)) == 0
    assert cnt.value == len(want_p)
    assert np.array_equal(m2.astype(bool), want_mask)
    assert np.array_equal(p2[:cnt.value], want_p) and np.array_equal(c2[:cnt.value], want_c)
    # NULL pair buffers are allowed; a NULL mask is refused
    assert lib.mtfhip_grid_fb_mask(n, a.ctypes.data, b.ctypes.data, f.ctypes.data, C.addressof(fbd), m2.ctypes.data, None, None, C.addressof(cnt)) == 0
    assert lib.mtfhip_grid_fb_mask(n, a.ctypes.data, b.ctypes.data, f.ctypes.data, C.addressof(fbd), None, None, None, C.addressof(cnt)) != 0


def _grid(oracle, frame, gs, ps, reset, fb_thresh, fb_reinit, est, n_model_pts=4):
    gp = oracle.GridParams(gs, gs, ps, ps, reset, 0, 1, fb_thresh, fb_reinit, n_model_pts)
    res = oracle.grid_res(gp)
    gssm = oracle.SSM(oracle.SSM_HOM, res[0], res[1])
    trks = []
    for _ in range(gs * gs):
        am = oracle.AM(oracle.AM_NCC, ps, ps)
        ssm = oracle.SSM(oracle.SSM_AFF, ps, ps)
        trks.append(oracle.Tracker(oracle.SM_ICLK, am, ssm, leven_marq=0, max_iters=20, epsilon=1e-4, hess_type=0))
    g = oracle.Grid(gssm, trks, grid_size=gs, patch_size=ps, reset_at_each_frame=reset, estimator=est, fb_err_thresh=fb_thresh, fb_reinit=fb_reinit,
                    n_model_pts=n_model_pts)
    g.set_image(frame)
    return g


@pytest.mark.parametrize("fb_reinit", [0, 1])
@pytest.mark.parametrize("reset", [1, 0])
def test_oracle_backward_estimation_round_trip(oracle, frame, fb_reinit, reset):
    """A rigid scene: every patch tracker that follows the motion forwards comes back to its starting centroid on the previous frame
    (fb_prev_pts ~ prev_pts, all kept) and the fit is the one the estimation-free tracker makes from the same pairs; then one patch's
    content is replaced in the new frame only -- that tracker's round trip misses and it is the one left out."""
    gs, ps = 4, 25
    est = least_squares_estimator(L.SSM_HOMOGRAPHY)
    region = synth.square_corners(256, 256, 240)
    f2 = synth.warp_frame(frame, synth.random_small_homography(np.random.default_rng(5), 0.15), (256.0, 256.0))
    g = _grid(oracle, frame, gs, ps, reset, 2.0, fb_reinit, est)
    plain = _grid(oracle, frame, gs, ps, reset, 0.0, fb_reinit, est)
    g.initialize(region); plain.initialize(region)
    start = g.prev_pts().copy()
    g.set_image(f2); plain.set_image(f2)
    g.update(); plain.update()
    assert g.fb_err_mask().all()
    np.testing.assert_allclose(g.fb_prev_pts(), start, rtol=0, atol=0.05)
    a, b = g.estimator_pairs()
    np.testing.assert_array_equal(a, start)
    np.testing.assert_array_equal(g.ssm_update(), plain.ssm_update())     # same pairs, same estimator
    np.testing.assert_array_equal(g.get_region(), plain.get_region())
    # the backward pass leaves the trackers where the forward pass put them (setRegion(tracker_location)) unless the frame's reset moved them
    if reset == 0:
        for t, loc in zip(g.trackers, g.fb_locations()):
            np.testing.assert_allclose(t.get_region(), loc, rtol=0, atol=1e-12)
    # --- a second pair of frames, patch 5's neighbourhood replaced by unrelated texture in the NEW frame only
    f3 = synth.warp_frame(f2, synth.random_small_homography(np.random.default_rng(6), 0.15), (256.0, 256.0))
    prev = g.prev_pts().copy()
    cx, cy = [int(round(v)) for v in prev[5]]
    other = synth.make_frame(512, 512, seed=99)
    f3 = f3.copy()
    f3[cy - 40:cy + 40, cx - 40:cx + 40] = other[100:180, 300:380]
    g.set_image(f3)
    g.update()
    mask = g.fb_err_mask()
    assert not mask[5] and mask.sum() >= gs * gs - 3
    a, b = g.estimator_pairs()
    assert len(a) == mask.sum()
    np.testing.assert_array_equal(a, prev[mask])


def test_oracle_backward_estimation_needs_the_grid_image(oracle, frame):
    g = _grid(oracle, frame, 2, 25, 1, 2.0, 1, least_squares_estimator(L.SSM_HOMOGRAPHY))
    h = oracle.Grid(g.ssm, g.trackers, grid_size=2, patch_size=25, estimator=least_squares_estimator(L.SSM_HOMOGRAPHY), fb_err_thresh=2.0)
    for t in h.trackers:
        t.am.set_curr_img(frame)       # the trackers have an image, the grid never saw one: no prev_img to go back to
    h.initialize(synth.square_corners(256, 256, 200))
    with pytest.raises(RuntimeError, match="set_image"):
        h.update()
