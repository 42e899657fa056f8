"""GridTracker (SM/src/GridTracker.cc) end to end on the device: the Python driver (mtf_amd.sm.GridTracker) and the C++ driver
(mtf::hip::Grid, libmtfhost.so) against the oracle's restatement of initialize / update / setRegion / resetTrackers over its own
per-patch nt:: trackers, frame after frame, in the three patch modes and the three reset modes, for grid SSMs of both kinds and
regions that are not parallelograms.  The estimator (estimateWarpFromPts: out of scope) is the same function on both sides."""
import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import host, synth
from mtf_amd.sm import GridTracker, least_squares_estimator

pytestmark = pytest.mark.gpu

CENTRE = (256.0, 256.0)
REGIONS = {
    "square": synth.square_corners(CENTRE[0], CENTRE[1], 280),
    "quad": synth.square_corners(CENTRE[0], CENTRE[1], 280) + np.array([[3.0, -2.0, 5.0, -4.0], [-1.5, 2.5, 4.0, -3.0]]),
    "projective": np.array([[130.0, 390, 370, 150], [125, 140, 385, 360]]),
}


def _frames(frame, n, seed):
    rng = np.random.default_rng(seed)
    out, cur = [], frame
    for _ in range(n):
        cur = synth.warp_frame(cur, synth.random_small_homography(rng, 0.15), CENTRE)
        out.append(cur)
    return out


def _oracle_grid(oracle, frame, grid_ssm, patch_am, patch_ssm, patch_sm, gs, ps, mode, reset, est, max_iters=20, hess_type=0):
    dyn, inside = mode
    gp = oracle.GridParams(gs, gs, ps, ps, reset, dyn, inside)
    res = oracle.grid_res(gp)
    gssm = oracle.SSM(grid_ssm, res[0], res[1])
    ams, trks = [], []
    for _ in range(gs * gs):
        am = oracle.AM(patch_am, ps, ps); am.set_curr_img(frame)
        ssm = oracle.SSM(patch_ssm, ps, ps)
        trks.append(oracle.Tracker(patch_sm, am, ssm, leven_marq=0, max_iters=max_iters, epsilon=1e-4, hess_type=hess_type))
        ams.append(am)
    g = oracle.Grid(gssm, trks, grid_size=gs, patch_size=ps, reset_at_each_frame=reset, dyn_patch_size=dyn, patch_centroid_inside=inside, estimator=est)
    return g, ams


MODES = {"centroid_inside": (0, 1), "grid_points": (0, 0), "dyn_patch": (1, 0)}


# r06 (was 2e-3 px throughout): the cv::Point2f centroids to one float ulp of a coordinate of a few hundred pixels (values that agree to
# ~1e-7 px in double may round to neighbouring floats), the double-precision region and patch corners -- which have been through the
# all-points fit of those floats -- to 1e-4 px; measured: <= 1.6e-5 px and <= 6e-6 px (profiles/r06_parity_record.jsonl)
PT_ULP, REGION_ATOL = 4e-5, 1e-4


def _compare(o, patch_corners, prev_pts, curr_pts, region, step):
    np.testing.assert_allclose(patch_corners, o.patch_corners(), rtol=0, atol=REGION_ATOL, err_msg="patch corners, frame %d" % step)
    np.testing.assert_allclose(prev_pts, o.prev_pts(), rtol=0, atol=PT_ULP, err_msg="prev_pts, frame %d" % step)
    if curr_pts is not None:
        np.testing.assert_allclose(curr_pts, o.curr_pts(), rtol=0, atol=PT_ULP, err_msg="curr_pts, frame %d" % step)
    np.testing.assert_allclose(region, o.get_region(), rtol=0, atol=REGION_ATOL, err_msg="region, frame %d" % step)


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("reset", [1, 2, 0])
@pytest.mark.parametrize("region,grid_ssm", [("square", L.SSM_HOMOGRAPHY), ("quad", L.SSM_HOMOGRAPHY), ("projective", L.SSM_HOMOGRAPHY), ("quad", L.SSM_AFFINE)])
def test_python_grid_tracker_follows_oracle(oracle, gpu_ctx, frame, mode, reset, region, grid_ssm):
    gs, ps = 5, 25
    est = least_squares_estimator(grid_ssm)
    frames = _frames(frame, 3, 77)
    o, ams = _oracle_grid(oracle, frame, grid_ssm, oracle.AM_NCC, oracle.SSM_AFF, oracle.SM_ICLK, gs, ps, MODES[mode], reset, est)
    gpu_ctx.set_image(frame)
    g = GridTracker(gpu_ctx, grid_size=gs, patch_size=ps, am=L.AM_NCC, ssm=L.SSM_AFFINE, max_iters=20, epsilon=1e-4, reset_at_each_frame=reset,
                    dyn_patch_size=MODES[mode][0], patch_centroid_inside=MODES[mode][1], grid_ssm=grid_ssm, estimator=est)
    assert g.res() == oracle.grid_res(o.gp)
    g.initialize(REGIONS[region]); o.initialize(REGIONS[region])
    # the layout itself to 1e-9 px (the verdict's bar), then the tracked quantities
    np.testing.assert_allclose(g.patch_corners(REGIONS[region]), o.patch_corners(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(g.grid_pts(REGIONS[region]), o.ssm.get("curr_pts").reshape(-1, 2), rtol=0, atol=1e-9)
    _compare(o, g.patch_corners(g.get_region()), g.prev_pts, g.curr_pts, g.get_region(), 0)
    for k, f in enumerate(frames):
        gpu_ctx.set_image(f)
        for am in ams:
            am.set_curr_img(f)
        g.update(); o.update()
        np.testing.assert_allclose(g.ssm_update, o.ssm_update(), rtol=0, atol=2e-5)
        _compare(o, g.patch_corners(g.get_region()) if reset else o.patch_corners(), g.prev_pts, g.curr_pts, g.get_region(), k + 1)
    # setRegion between frames (GridTracker.cc:287-292)
    moved = g.get_region() + np.array([[1.5], [-0.75]])
    g.set_region(moved); o.set_region(moved)
    gpu_ctx.set_image(frames[0])
    for am in ams:
        am.set_curr_img(frames[0])
    g.update(); o.update()
    _compare(o, g.patch_corners(g.get_region()) if reset else o.patch_corners(), g.prev_pts, g.curr_pts, g.get_region(), 9)
    g.tracker.batch.close()


@pytest.mark.parametrize("patch_sm,patch_am,patch_ssm,hess", [(L.SM_FCLK, L.AM_SSD, L.SSM_HOMOGRAPHY, 1), (L.SM_ESM, L.AM_NCC, L.SSM_AFFINE, 2)])
def test_python_grid_tracker_other_patch_trackers(oracle, gpu_ctx, frame, patch_sm, patch_am, patch_ssm, hess):
    """grid_sm / grid_am / grid_ssm are free in the reference (mtf.h:777-801): FCLK + SSD + homography and ESM + NCC + affine patches,
    which take the launch-per-pass device loop instead of the one-launch ICLK kernel"""
    gs, ps = 3, 30
    est = least_squares_estimator(L.SSM_HOMOGRAPHY)
    frames = _frames(frame, 2, 78)
    o, ams = _oracle_grid(oracle, frame, L.SSM_HOMOGRAPHY, patch_am, patch_ssm, patch_sm, gs, ps, (0, 1), 1, est, hess_type=hess)
    gpu_ctx.set_image(frame)
    g = GridTracker(gpu_ctx, grid_size=gs, patch_size=ps, am=patch_am, ssm=patch_ssm, sm=patch_sm, max_iters=20, epsilon=1e-4, hess_type=hess, estimator=est)
    g.initialize(REGIONS["quad"]); o.initialize(REGIONS["quad"])
    for k, f in enumerate(frames):
        gpu_ctx.set_image(f)
        for am in ams:
            am.set_curr_img(f)
        g.update(); o.update()
        _compare(o, g.patch_corners(g.get_region()), g.prev_pts, g.curr_pts, g.get_region(), k + 1)
    g.tracker.batch.close()


@pytest.mark.parametrize("mode,reset,region,grid_ssm", [("centroid_inside", 1, "quad", L.SSM_HOMOGRAPHY), ("centroid_inside", 2, "projective", L.SSM_HOMOGRAPHY),
                                                        ("grid_points", 1, "square", L.SSM_AFFINE), ("dyn_patch", 0, "quad", L.SSM_HOMOGRAPHY),
                                                        ("dyn_patch", 1, "projective", L.SSM_AFFINE)])
def test_cpp_grid_driver_follows_oracle(oracle, frame, mode, reset, region, grid_ssm):
    """mtf::hip::Grid (libmtfhost.so) with the reference's parameter block: setImage / initialize / update / setRegion / getRegion"""
    gs, ps = 5, 25
    est = least_squares_estimator(grid_ssm)
    frames = _frames(frame, 3, 79)
    o, ams = _oracle_grid(oracle, frame, grid_ssm, oracle.AM_NCC, oracle.SSM_AFF, oracle.SM_ICLK, gs, ps, MODES[mode], reset, est)
    g = host.CppGridTracker(grid_size=gs, patch_size=ps, patch_sm=L.SM_ICLK, patch_am=L.AM_NCC, patch_ssm=L.SSM_AFFINE, grid_ssm=grid_ssm,
                            reset_at_each_frame=reset, dyn_patch_size=MODES[mode][0], patch_centroid_inside=MODES[mode][1], max_iters=20, epsilon=1e-4,
                            hess_type=0, estimator=est)
    g.set_image(frame)
    g.initialize(REGIONS[region]); o.initialize(REGIONS[region])
    np.testing.assert_allclose(g.patch_corners(), o.patch_corners(), rtol=0, atol=1e-9)
    _compare(o, g.patch_corners(), g.prev_pts(), g.curr_pts(), g.get_region(), 0)
    for k, f in enumerate(frames):
        g.set_image(f)
        for am in ams:
            am.set_curr_img(f)
        g.update(); o.update()
        np.testing.assert_allclose(g.ssm_update(), o.ssm_update(), rtol=0, atol=2e-5)
        _compare(o, g.patch_corners() if reset else o.patch_corners(), g.prev_pts(), g.curr_pts(), g.get_region(), k + 1)
    moved = g.get_region() + np.array([[-2.0], [1.25]])
    g.set_region(moved); o.set_region(moved)
    g.set_image(frames[1])
    for am in ams:
        am.set_curr_img(frames[1])
    g.update(); o.update()
    _compare(o, g.patch_corners() if reset else o.patch_corners(), g.prev_pts(), g.curr_pts(), g.get_region(), 9)


def test_cpp_grid_driver_builtin_estimator_and_refusals(frame):
    """the built-in all-points least-squares fit (normal equations in the normalised frame) against the NumPy one (SVD), and the
    constructor's refusals (GridTracker.cc:124-134 equivalents)"""
    g = host.CppGridTracker(grid_size=6, patch_size=25, max_iters=15, hess_type=0)
    g.set_image(frame)
    g.initialize(REGIONS["quad"])
    f2 = _frames(frame, 1, 80)[0]
    g.set_image(f2)
    g.update()
    assert np.isfinite(g.get_region()).all()
    # (with reset_at_each_frame = 1 prev_pts are replaced by the reset: the fit is checked on instances that do not reset)
    h = host.CppGridTracker(grid_size=6, patch_size=25, max_iters=15, hess_type=0, reset_at_each_frame=0)
    h.set_image(frame); h.initialize(REGIONS["quad"])
    prev = h.prev_pts().copy()
    h.set_image(f2); h.update()
    # (h22 = 1 against |h| = 1: two algebraic least-squares problems that agree on the noise-free fit, so they are compared through
    # what they do to the region's corners, not entry by entry)
    from mtf_amd.api import apply_warp_to_pts
    want = least_squares_estimator(L.SSM_HOMOGRAPHY)(prev, h.curr_pts())
    np.testing.assert_allclose(apply_warp_to_pts(L.SSM_HOMOGRAPHY, REGIONS["quad"], h.ssm_update()),
                               apply_warp_to_pts(L.SSM_HOMOGRAPHY, REGIONS["quad"], want), rtol=0, atol=1e-3)
    a = host.CppGridTracker(grid_size=6, patch_size=25, max_iters=15, hess_type=0, reset_at_each_frame=0, grid_ssm=L.SSM_AFFINE)
    a.set_image(frame); a.initialize(REGIONS["quad"])
    prev = a.prev_pts().copy()
    a.set_image(f2); a.update()
    np.testing.assert_allclose(a.ssm_update(), least_squares_estimator(L.SSM_AFFINE)(prev, a.curr_pts()), rtol=0, atol=1e-8)
    with pytest.raises(host.HostError):
        host.CppGridTracker(grid_size=0, patch_size=25)
    with pytest.raises(host.HostError, match="before initialize"):
        host.CppGridTracker(grid_size=2, patch_size=25).update()


def test_grid_frame_abi_refuses_a_mismatched_batch(gpu_ctx, frame):
    gpu_ctx.set_image(frame)
    g = GridTracker(gpu_ctx, grid_size=3, patch_size=25, max_iters=5)
    g.initialize(REGIONS["square"])
    wrong = L.GridDesc(4, 4, 25, 25, 1, 0, 1)
    with pytest.raises(mtf_amd.MtfHipError, match="mismatch between the grid dimensions"):
        g.tracker.batch.grid_frame(wrong, g.tracker.sm, REGIONS["square"])
    g.tracker.batch.close()


@pytest.mark.parametrize("am,ssm,ps", [(L.AM_NCC, L.SSM_AFFINE, 25), (L.AM_SSD, L.SSM_AFFINE, 25), (L.AM_NCC, L.SSM_HOMOGRAPHY, 30), (L.AM_SSD, L.SSM_HOMOGRAPHY, 16),
                                       (L.AM_NCC, L.SSM_AFFINE, 32)])
def test_fused_template_init_equals_call_by_call(oracle, gpu_ctx, frame, am, ssm, ps, monkeypatch):
    """r05: nt::ICLK::initialize of small patches in ONE launch (kernels_init.hip, init_template's fast path) against the call-by-call
    form (MTFHIP_INIT_FUSED=0: ~15 launches, six host round trips): the per-pixel arrays -- I0, It, dI0_dx, J0 -- are the same BITS; the
    reduced quantities (constant self Hessian, its inverse as the tracking launch uses it, NCC scalars) agree to rounding; the frames
    tracked afterwards land on the same corners; and the host mirrors are folded in lazily (the interface getters see them)."""
    gs = 4
    f2 = _frames(frame, 1, 81)[0]
    region = REGIONS["quad"]
    out = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("MTFHIP_INIT_FUSED", fused)
        monkeypatch.setenv("MTFHIP_GRID_FUSED", fused)   # (the fused reset also lays the grid out in the same launch)
        gpu_ctx.set_image(frame)
        g = GridTracker(gpu_ctx, grid_size=gs, patch_size=ps, am=am, ssm=ssm, max_iters=15, epsilon=1e-5, reset_at_each_frame=1)
        g.initialize(region)
        b = g.tracker.batch
        arrays = [b.read(L.BUF_I0).copy(), b.read(L.BUF_IT).copy(), b.read(L.BUF_DI0_DX).copy(), b.read(L.BUF_J0).copy(), b.read(L.BUF_INIT_PTS).copy(),
                  b.read(L.BUF_CURR_PTS).copy(), b.get_state().copy(), b.get_corners().copy()]
        H0 = b.cmpt_self_hessian(L.BUF_J0).copy() if am == L.AM_SSD else None     # through the interface: the host mirrors must be current
        gpu_ctx.set_image(f2)
        c1, cen1 = g.update_patches()
        n1 = g.n_iters.copy()
        r2 = g.update()            # a whole frame: track, fit, re-initialise on the new grid (fused or not), ...
        c3, _ = g.update_patches() # ... and track from the re-initialised templates
        r4 = g.update(); r5 = g.update()   # two more frames: each re-initialisation supersedes the record of the one before (never folded in) ...
        # ... and the getters that read the host mirrors still see the LAST template's (the pending record is folded in when they ask)
        H5 = b.cmpt_self_hessian(L.BUF_J0).copy() if am == L.AM_SSD else None
        f5 = np.asarray(b.get_similarity()).copy()
        out[fused] = (arrays, H0, c1.copy(), n1, r2.copy(), c3.copy(), r4.copy(), r5.copy(), H5, f5)
        b.close()
    for x, y, what in zip(out["0"][0], out["1"][0], ("I0", "It", "dI0_dx", "J0", "init_pts", "curr_pts", "state", "corners")):
        assert np.array_equal(x, y), what
    if out["0"][1] is not None:
        np.testing.assert_allclose(out["1"][1], out["0"][1], rtol=1e-12)
    assert np.array_equal(out["0"][3], out["1"][3])
    np.testing.assert_allclose(out["1"][2], out["0"][2], rtol=0, atol=1e-8)
    np.testing.assert_allclose(out["1"][4], out["0"][4], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["1"][5], out["0"][5], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["1"][6], out["0"][6], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out["1"][7], out["0"][7], rtol=0, atol=1e-5)
    if out["0"][8] is not None:
        np.testing.assert_allclose(out["1"][8], out["0"][8], rtol=1e-7)   # (the two forms sit on regions that agree to ~1e-6 px by now)
    np.testing.assert_allclose(out["1"][9], out["0"][9], rtol=1e-7, atol=1e-9)
    # and against the oracle's own initialize + update of one patch
    gpu_ctx.set_image(frame)
    g = GridTracker(gpu_ctx, grid_size=gs, patch_size=ps, am=am, ssm=ssm, max_iters=15, epsilon=1e-5)
    g.initialize(region)
    patches = g.patch_corners(region)
    gpu_ctx.set_image(f2)
    c, _ = g.update_patches()
    for t in (0, 7, 15):
        o_ssm = oracle.SSM(ssm, ps, ps); o_am = oracle.AM(am, ps, ps); o_am.set_curr_img(frame)
        trk = oracle.Tracker(L.SM_ICLK, o_am, o_ssm, leven_marq=0, max_iters=15, epsilon=1e-5, hess_type=0)
        trk.initialize(patches[t]); o_am.set_curr_img(f2); trk.update()
        np.testing.assert_allclose(c[t], trk.get_region(), rtol=0, atol=5e-4)
    g.tracker.batch.close()


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("am,region", [(L.AM_NCC, "projective"), (L.AM_SSD, "quad"), (L.AM_NCC, "square")])
def test_grid_frame_patches_laid_out_by_the_kernel_equal_the_host_layout(gpu_ctx, frame, mode, am, region, monkeypatch):
    """r05: with fixed-size patches mtfhip_grid_frame hands the kernel the GRID's region and every workgroup computes its own patch
    corners (grid_patch_corners_hd, the expressions of mtfhip_grid_layout) -- no PCIe read in front of the grid layout, and the host
    lays the patches out for its mirrors behind the launch.  MTFHIP_GRID_LAYOUT_DEV=0 keeps the host layout in front: the same
    iteration counts, corners, centroids, template grids, states and patch regions bit for bit, in both arithmetic modes, frame after
    frame; dyn_patch_size patches always take the host layout."""
    dyn, inside = MODES[mode]
    reg = REGIONS[region]
    frames = _frames(frame, 3, 21)
    rec, reinit = {}, {}
    for dev in ("1", "0"):
        monkeypatch.setenv("MTFHIP_GRID_LAYOUT_DEV", dev)
        gpu_ctx.set_image(frame)
        gt = GridTracker(gpu_ctx, grid_size=6, patch_size=21, am=am, ssm=L.SSM_AFFINE, grid_ssm=L.SSM_HOMOGRAPHY, max_iters=8, epsilon=1e-4,
                         dyn_patch_size=dyn, patch_centroid_inside=inside, reset_at_each_frame=2)
        gt.initialize(reg)
        b = gt.tracker.batch
        out = []
        for k, f in enumerate(frames):
            gpu_ctx.set_image(f)
            b.set_math_mode(mtf_amd.MATH_FAST if k != 1 else mtf_amd.MATH_REPLAY)
            n, c, m = b.grid_frame(gt.gd, gt.tracker.sm, reg + 0.3 * k)
            out.append([n.copy(), c.copy(), m.copy(), b.read(L.BUF_INIT_PTS).copy(), b.read(L.BUF_INIT_HXY).copy(), b.read(L.BUF_INIT_Z).copy(),
                        b.get_state().copy(), b.get_corners().copy()])
            # ... and what the host mirrors hold afterwards: a plain update() from there
            n2, c2, m2 = b.grid_frame(gt.gd, gt.tracker.sm, None)
            out[-1] += [n2.copy(), c2.copy(), m2.copy()]
        # resetTrackers(reinit = true) on the last frame: k_template_init lays its patches out itself in the same way
        b.set_math_mode(mtf_amd.MATH_FAST)
        pcs, pp = b.grid_reset(gt.gd, gt.tracker.sm, reg + 1.0, True)
        n3, c3, m3 = b.grid_frame(gt.gd, gt.tracker.sm, None)
        reinit[dev] = [pcs.copy(), pp.copy(), b.read(L.BUF_I0).copy(), b.read(L.BUF_J0).copy(), b.read(L.BUF_INIT_PTS).copy(), b.get_corners().copy(),
                       n3.copy(), c3.copy(), m3.copy()]
        rec[dev] = out
        del gt
    for x, y, what in zip(reinit["1"], reinit["0"], ("patch corners", "prev_pts", "I0", "J0", "init_pts", "corner mirrors", "n_iters", "corners", "centroids")):
        assert np.array_equal(x, y), ("reinit", what)
    for k in range(len(frames)):
        for x, y, what in zip(rec["1"][k], rec["0"][k], ("n_iters", "corners", "centroids", "init_pts", "init_hxy", "init_z", "state", "corner mirrors",
                                                       "n_iters of the next update", "corners of the next update", "centroids of the next update")):
            assert np.array_equal(x, y), (k, what)
    assert all((r[0] > 0).all() for r in rec["1"])


# ------------------------------------------------------------------------------------------------------------------------------
# r06: forward-backward error estimation (SM/src/GridTracker.cc:186-190, 241-243, 263-266, 294-343; shipped grid_fb_err_thresh 2,
# grid_fb_reinit 1) and the per-patch comparison of the grid's tracked quantities at the north-star tolerance
# ------------------------------------------------------------------------------------------------------------------------------
def _spoil(img, centre):
    other = synth.make_frame(512, 512, seed=99)
    cx, cy = [int(round(v)) for v in centre]
    out = img.copy()
    out[cy - 40:cy + 40, cx - 40:cx + 40] = other[100:180, 300:380]
    return out


@pytest.mark.parametrize("fb_reinit", [0, 1])
@pytest.mark.parametrize("reset", [1, 0, 2])
@pytest.mark.parametrize("fb_err_thresh", [0.0, 2.0])
@pytest.mark.parametrize("driver", ["python", "cpp"])
def test_grid_forward_backward_follows_oracle(oracle, gpu_ctx, frame, parity_record, driver, fb_err_thresh, reset, fb_reinit):
    """GridTracker::update with and without backwardEstimation, Python and C++ drivers: fb_err_mask bit for bit, fb_prev_pts / curr_pts /
    prev_pts as the cv::Point2f the reference holds (one float ulp of a few hundred pixels = 3e-5 allowed for the rounding of values that
    agree to ~1e-7 px in double), the double-precision patch regions at 1e-6 px, the state update of the fit and the region."""
    if fb_err_thresh == 0.0 and fb_reinit == 0:
        pytest.skip("fb_reinit is not read with the estimation off")
    gs, ps, grid_ssm = 5, 25, L.SSM_HOMOGRAPHY
    est = least_squares_estimator(grid_ssm)
    region = REGIONS["quad"]
    gp = oracle.GridParams(gs, gs, ps, ps, reset, 0, 1, fb_err_thresh, fb_reinit, 4)
    res = oracle.grid_res(gp)
    gssm = oracle.SSM(grid_ssm, res[0], res[1])
    trks = []
    for _ in range(gs * gs):
        trks.append(oracle.Tracker(oracle.SM_ICLK, oracle.AM(oracle.AM_NCC, ps, ps), oracle.SSM(oracle.SSM_AFF, ps, ps), leven_marq=0, max_iters=20, epsilon=1e-4,
                                   hess_type=0))
    o = oracle.Grid(gssm, trks, grid_size=gs, patch_size=ps, reset_at_each_frame=reset, estimator=est, fb_err_thresh=fb_err_thresh, fb_reinit=fb_reinit, n_model_pts=4)
    kw = dict(grid_size=gs, patch_size=ps, max_iters=20, epsilon=1e-4, reset_at_each_frame=reset, grid_ssm=grid_ssm, estimator=est, fb_err_thresh=fb_err_thresh,
              fb_reinit=fb_reinit, n_model_pts=4)
    if driver == "python":
        g = GridTracker(gpu_ctx, am=L.AM_NCC, ssm=L.SSM_AFFINE, **kw)
        set_image = gpu_ctx.set_image
        get = dict(prev=lambda: g.prev_pts, curr=lambda: g.curr_pts, fb=lambda: g.fb_prev_pts, mask=lambda: g.fb_err_mask, upd=lambda: g.ssm_update)
    else:
        g = host.CppGridTracker(patch_sm=L.SM_ICLK, patch_am=L.AM_NCC, patch_ssm=L.SSM_AFFINE, hess_type=0, **kw)
        set_image = g.set_image
        get = dict(prev=g.prev_pts, curr=g.curr_pts, fb=g.fb_prev_pts, mask=g.fb_err_mask, upd=g.ssm_update)
    set_image(frame); o.set_image(frame)
    g.initialize(region); o.initialize(region)
    frames = _frames(frame, 3, 91)
    ULP = 4e-5
    worst = dict(fb=0.0, curr=0.0, upd=0.0, region=0.0)
    for k, f in enumerate(frames):
        if k == 2:
            f = _spoil(f, o.prev_pts()[7])
        set_image(f); o.set_image(f)
        g.update(); o.update()
        if fb_err_thresh > 0:
            assert np.array_equal(get["mask"](), o.fb_err_mask()), "fb_err_mask, frame %d" % k
            np.testing.assert_allclose(get["fb"](), o.fb_prev_pts(), rtol=0, atol=ULP, err_msg="fb_prev_pts, frame %d" % k)
            worst["fb"] = max(worst["fb"], float(np.abs(get["fb"]() - o.fb_prev_pts()).max()))
            if k == 2:
                assert not o.fb_err_mask()[7] and o.fb_err_mask().sum() >= gs * gs - 4
        np.testing.assert_allclose(get["curr"](), o.curr_pts(), rtol=0, atol=ULP, err_msg="curr_pts, frame %d" % k)
        np.testing.assert_allclose(get["prev"](), o.prev_pts(), rtol=0, atol=ULP, err_msg="prev_pts, frame %d" % k)
        # (the fit sees cv::Point2f centroids: values that agree to ~1e-7 px in double can round to neighbouring floats, 3e-5 px apart at a few
        # hundred pixels, and the all-points fit passes that on to its translation entries at about the same size)
        np.testing.assert_allclose(get["upd"](), o.ssm_update(), rtol=0, atol=2e-5, err_msg="ssm_update, frame %d" % k)
        np.testing.assert_allclose(g.get_region(), o.get_region(), rtol=0, atol=2e-4, err_msg="region, frame %d" % k)
        worst["curr"] = max(worst["curr"], float(np.abs(get["curr"]() - o.curr_pts()).max()))
        worst["upd"] = max(worst["upd"], float(np.abs(get["upd"]() - o.ssm_update()).max()))
        worst["region"] = max(worst["region"], float(np.abs(g.get_region() - o.get_region()).max()))
    parity_record.append(dict(test="grid_forward_backward", driver=driver, fb_err_thresh=fb_err_thresh, fb_reinit=fb_reinit, reset=reset, **worst))
    if driver == "python":
        g.tracker.batch.close()


@pytest.mark.parametrize("fb_reinit", [0, 1])
def test_grid_backward_abi_patch_regions_follow_oracle(oracle, frame, fb_reinit):
    """mtfhip_grid_backward alone, in double precision (before the cv::Point2f rounding): where every patch tracker arrives on the previous
    frame against the oracle's per-patch trackers at 1e-6 px, the iteration counts, that the current image is the current one again
    afterwards and the trackers sit on their forward locations (setRegion(tracker_location)); and the refusals."""
    gs, ps = 4, 25
    est = least_squares_estimator(L.SSM_HOMOGRAPHY)
    f2 = _frames(frame, 1, 92)[0]
    gpu_ctx = mtf_amd.Context(0)     # (its own context: the session's has kept a previous image in the tests above)
    gpu_ctx.set_image(frame)
    g = GridTracker(gpu_ctx, grid_size=gs, patch_size=ps, am=L.AM_NCC, ssm=L.SSM_AFFINE, max_iters=25, epsilon=1e-6, reset_at_each_frame=0, estimator=est)
    b = g.tracker.batch
    fbd = L.GridFbDesc(2.0, fb_reinit, 4)
    g.initialize(REGIONS["square"])
    with pytest.raises(mtf_amd.MtfHipError, match="no previous image"):
        b.grid_backward(g.gd, g.tracker.sm, fbd)
    gpu_ctx.keep_prev()
    gpu_ctx.set_image(f2)
    n_fwd, fwd, cen = b.grid_frame(g.gd, g.tracker.sm, None)
    fwd = fwd.copy()
    n_back, back, fb_pts = b.grid_backward(g.gd, g.tracker.sm, fbd)
    np.testing.assert_allclose(b.get_corners(), fwd, rtol=0, atol=1e-12)     # setRegion(tracker_location)
    patches = g.patch_corners(REGIONS["square"])
    worst = 0.0
    for t in range(gs * gs):
        am, ssm = oracle.AM(oracle.AM_NCC, ps, ps), oracle.SSM(oracle.SSM_AFF, ps, ps)
        am.set_curr_img(frame)
        trk = oracle.Tracker(oracle.SM_ICLK, am, ssm, leven_marq=0, max_iters=25, epsilon=1e-6, hess_type=0)
        trk.initialize(patches[t]); am.set_curr_img(f2); trk.update()
        loc = trk.get_region().copy()
        np.testing.assert_allclose(fwd[t], loc, rtol=0, atol=1e-6)
        if fb_reinit:
            trk.initialize(loc)
        am.set_curr_img(frame)
        n_o = trk.update()
        np.testing.assert_allclose(back[t], trk.get_region(), rtol=0, atol=1e-6, err_msg="patch %d" % t)
        worst = max(worst, float(np.abs(back[t] - trk.get_region()).max()))
        assert n_back[t] == n_o
        np.testing.assert_allclose(fb_pts[t], trk.get_region().mean(axis=1).astype(np.float32), rtol=0, atol=4e-5)
    # the image is the current one again: a plain frame from here equals the oracle's next forward step on f2 from `fwd`
    assert gpu_ctx.has_prev()
    wrong = L.GridDesc(3, 3, ps, ps, 0, 0, 1)
    with pytest.raises(mtf_amd.MtfHipError, match="mismatch between the grid dimensions"):
        b.grid_backward(wrong, g.tracker.sm, fbd)
    b.close()
    gpu_ctx.close()


@pytest.mark.parametrize("am,ssm,ps", [(L.AM_NCC, L.SSM_AFFINE, 25), (L.AM_SSD, L.SSM_AFFINE, 25), (L.AM_NCC, L.SSM_HOMOGRAPHY, 30)])
def test_grid_template_init_and_first_iteration_against_the_oracles_trackers(oracle, gpu_ctx, frame, parity_record, am, ssm, ps):
    """r05's verdict: k_template_init and the one-launch grid loop were checked against the call-by-call HIP form only.  Here against the
    oracle's nt::ICLK::initialize arrays directly -- I0, dI0_dx, J0 / H0 to the rounding of the sample points -- and, per patch, g, the state
    update and the corners of every iteration of the first frame at the north-star tolerance (1e-5 relative), the tracked corners at
    1e-6 px in double (before any cv::Point2f rounding)."""
    gs = 4
    f2 = _frames(frame, 1, 93)[0]
    region = REGIONS["quad"]
    gpu_ctx.set_image(frame)
    g = GridTracker(gpu_ctx, grid_size=gs, patch_size=ps, am=am, ssm=ssm, max_iters=12, epsilon=1e-6, reset_at_each_frame=1)
    g.initialize(region)                      # resetTrackers(true): k_template_init in region / layout mode
    b = g.tracker.batch
    patches = g.patch_corners(region)
    I0, dI0, J0 = b.read(L.BUF_I0).copy(), b.read(L.BUF_DI0_DX).copy(), b.read(L.BUF_J0).copy()
    H0 = b.cmpt_self_hessian(L.BUF_J0).copy() if am == L.AM_SSD else None
    b.track_trace(12)
    gpu_ctx.set_image(f2)
    n, corners, _ = b.grid_frame(g.gd, g.tracker.sm, None)
    n, corners = n.copy(), corners.copy()
    trace = b.read_track_trace(n)
    S = b.S
    worst = dict(J0=0.0, H0=0.0, g=0.0, dp=0.0, corners=0.0)
    for t in range(gs * gs):
        o_am, o_ssm = oracle.AM(am, ps, ps), oracle.SSM(ssm, ps, ps)
        o_am.set_curr_img(frame)
        trk = oracle.Tracker(oracle.SM_ICLK, o_am, o_ssm, leven_marq=0, max_iters=12, epsilon=1e-6, hess_type=0)
        trk.initialize(patches[t])
        # (the sample points themselves agree to ~1e-12 px -- closed-form square-to-quadrilateral map against the oracle's SVD DLT -- so the
        # samples are compared at that distance times the image gradient, not bit for bit; the finite difference over 2e-8 px amplifies it)
        np.testing.assert_allclose(I0[t], o_am.get("I0"), rtol=0, atol=1e-9, err_msg="I0 of patch %d" % t)
        np.testing.assert_allclose(dI0[t], o_am.get("dI0_dx").reshape(2, -1).T, rtol=0, atol=2e-5, err_msg="dI0_dx of patch %d" % t)
        # J0: the oracle's cmptWarpedPixJacobian of the DEVICE's gradient (the formula, free of the finite difference's amplification) ...
        J0_o = o_ssm.cmpt_warped_pix_jacobian(np.ascontiguousarray(dI0[t].T).ravel()).reshape(S, -1).T
        np.testing.assert_allclose(J0[t], J0_o, rtol=1e-11, atol=1e-9, err_msg="J0 of patch %d" % t)
        worst["J0"] = max(worst["J0"], float(np.abs(J0[t] - J0_o).max() / np.abs(J0_o).max()))
        o_am.set_curr_img(f2)
        n_o = trk.update()
        rec = trk.trace()
        assert n[t] == n_o == len(rec)
        if H0 is not None:
            e = np.linalg.norm(H0[t] - rec[0]["H"]) / np.linalg.norm(rec[0]["H"])
            assert e < 1e-7, "H0 of patch %d: %.3e" % (t, e)
            worst["H0"] = max(worst["H0"], float(e))
        g0, d0 = np.linalg.norm(rec[0]["g"]), np.linalg.norm(rec[0]["dp"])
        for k in range(n_o):
            # north-star tolerance (1e-5 relative) on the first iteration; the later ones -- g and the update shrink towards zero as the
            # patch converges, the reference's own finite-difference noise floor does not -- on the first iteration's scale, as
            # _fused_follow holds g to its Cauchy-Schwarz scale
            eg = np.linalg.norm(trace[t][k]["g"] - rec[k]["g"]) / (np.linalg.norm(rec[k]["g"]) if k == 0 else g0)
            ed = np.linalg.norm(trace[t][k]["dp"] - rec[k]["dp"]) / (np.linalg.norm(rec[k]["dp"]) if k == 0 else d0)
            ec = np.abs(trace[t][k]["corners"] - rec[k]["corners"]).max()
            assert eg < 1e-5 and ed < 1e-5, "patch %d iteration %d: g %.3e dp %.3e" % (t, k, eg, ed)
            worst["g"], worst["dp"] = max(worst["g"], float(eg)), max(worst["dp"], float(ed))
            assert ec < 1e-6, "patch %d iteration %d: corners %.3e px" % (t, k, ec)
            worst["corners"] = max(worst["corners"], float(ec))
        np.testing.assert_allclose(corners[t], trk.get_region(), rtol=0, atol=1e-6)
    parity_record.append(dict(test="grid_template_init_and_first_iteration", am=int(am), ssm=int(ssm), patch=ps, **worst))
    b.track_trace(0)
    b.close()


@pytest.mark.parametrize("reset,fb_reinit", [(1, 1), (1, 0), (0, 0)], ids=["reset1-reinit1", "reset1-reinit0", "reset0-reinit0"])
@pytest.mark.parametrize("where", ["centre", "border"])
@pytest.mark.parametrize("ps", [16, 25, 32])
@pytest.mark.parametrize("am", [L.AM_NCC, L.AM_SSD], ids=["ncc", "ssd"])
def test_grid_fb_one_launch_equals_three(frame, am, ps, where, reset, fb_reinit, monkeypatch):
    """the shipped configuration's frame (reset_at_each_frame 1, fb_err_thresh 2, fb_reinit 1) as ONE launch (k_grid_fb: a patch's update(),
    initialize(tracker_location) and update() on the previous frame in its workgroup) against the three launches with two host waits
    (k_iclk_track, k_template_init, k_iclk_track; MTFHIP_GRID_FB_FUSED=0): the same expressions in the same order, so the same bits --
    iteration counts, patch regions, centroids, fb_prev_pts, the mask, the fitted update and the grid's region -- over four frames, the
    third of them with a spoiled patch (a backward pass that does not come home).  1, 3 and 4 pixels per thread."""
    gs = 5
    est = least_squares_estimator(L.SSM_HOMOGRAPHY)
    kw = dict(grid_size=gs, patch_size=ps, max_iters=12, epsilon=1e-4, reset_at_each_frame=reset, grid_ssm=L.SSM_HOMOGRAPHY, estimator=est, fb_err_thresh=2.0, fb_reinit=fb_reinit,
              n_model_pts=4, am=am, ssm=L.SSM_AFFINE)
    ctxs = [mtf_amd.Context(0), mtf_amd.Context(0)]
    gts = [GridTracker(c, **kw) for c in ctxs]
    forms = ["1", "0"]
    for c, g, f in zip(ctxs, gts, forms):
        monkeypatch.setenv("MTFHIP_GRID_FB_FUSED", f)
        # "border": the grid reaches past the frame's right and lower edges -- patches whose samples take the border value 128, whose texel
        # windows are clamped into the frame or unusable (the global sampling path), templates re-initialised partly outside the frame
        region = REGIONS["quad"] if where == "centre" else REGIONS["quad"] + np.array([[118.0], [112.0]])
        c.set_image(frame); g.initialize(region)
    frames = _frames(frame, 4, 17)
    for k, fr in enumerate(frames):
        if k == 2:
            fr = _spoil(fr, gts[0].prev_pts[7])
        rec = []
        for c, g, f in zip(ctxs, gts, forms):
            monkeypatch.setenv("MTFHIP_GRID_FB_FUSED", f)
            c.set_image(fr); g.update()
            rec.append(dict(curr=g.curr_pts.copy(), fb=g.fb_prev_pts.copy(), mask=g.fb_err_mask.copy(), upd=np.array(g.ssm_update), region=g.get_region().copy(),
                            n_iters=np.array(g.n_iters) if hasattr(g, "n_iters") else None, patches=np.array(g.patch_regions) if hasattr(g, "patch_regions") else None))
        a, b = rec
        for key in a:
            if a[key] is not None:
                # (equal_nan: a patch that has run away -- the spoiled one, at the border -- comes back from the previous frame as NaN in both forms)
                assert np.array_equal(a[key], b[key], equal_nan=a[key].dtype.kind == "f"), "%s differs between the one-launch and the three-launch frame, frame %d" % (key, k)
        if k == 2 and fb_reinit:     # (without the re-initialisation the backward pass carries the OLD template back: a spoiled patch may well come home)
            assert not a["mask"][7]
    for g in gts:
        g.tracker.batch.close()
    for c in ctxs:
        c.close()
