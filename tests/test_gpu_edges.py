"""GPU: edge cases of the domain -- tiny and ragged patches (N below one wave, not a multiple of the workgroup), regions
partly or wholly outside the frame (constant border 128: zero gradient, singular Hessian), degenerate corners, one
candidate, many targets with different fates in one batch."""
import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("resx,resy", [(5, 7), (7, 9), (16, 17), (33, 31)])
@pytest.mark.parametrize("sm_kind", [L.SM_ESM, L.SM_FCLK, L.SM_ICLK])
@pytest.mark.parametrize("am", [L.AM_NCC, L.AM_MI])
def test_ragged_patches_ncc_mi(oracle, gpu_ctx, frame, frame2, resx, resy, sm_kind, am):
    """The same sizes through the fused NCC kernel (72-wide rows, partial last row) and the fused MI iteration (64-pixel
    chunks of the histogram / Hessian kernels with a ragged tail, MFMA bin mode): first iteration against the oracle on
    the oracle's own sample grid, so only summation order differs."""
    corners = synth.square_corners(250, 244, 40)
    o_ssm = oracle.SSM(L.SSM_AFFINE, resx, resy); o_am = oracle.AM(am, resx, resy); o_am.set_curr_img(frame)
    trk = oracle.Tracker(sm_kind, o_am, o_ssm, leven_marq=0, max_iters=1, epsilon=-1.0)
    trk.initialize(corners)
    o_am.set_curr_img(frame2)
    trk.update()
    rec = trk.trace()[0]
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, am, L.SSM_AFFINE, resx, resy, 1)
    b.set_corners(corners[None])
    hm = o_ssm.get("init_pts_hm").reshape(-1, 3)
    b.write(L.BUF_INIT_PTS, o_ssm.get("init_pts").reshape(1, -1, 2).transpose(0, 2, 1))
    b.write(L.BUF_INIT_HXY, hm[:, :2].T[None])
    b.write(L.BUF_INIT_Z, hm[:, 2][None])
    b.set_state(np.zeros((1, b.S)))
    sm = mtf_amd.sm_desc(sm_kind, materialize=1, leven_marq=0)
    b.init_template(sm)
    gpu_ctx.set_image(frame2)
    f, g, H = b.iterate(sm)
    assert abs(f[0] - rec["f"]) <= 1e-9 * abs(rec["f"]) + 1e-12
    assert rel(H[0], rec["H"]) < 1e-8
    assert np.linalg.norm(g[0] - rec["g"]) <= 1e-8 * max(np.linalg.norm(rec["g"]), np.sqrt(abs(np.trace(rec["H"]))) * 1e-3)
    b.close()


@pytest.mark.parametrize("resx,resy", [(2, 2), (2, 3), (7, 9), (16, 17), (33, 31)])
@pytest.mark.parametrize("sm_kind", [L.SM_ESM, L.SM_FCLK, L.SM_ICLK])
def test_tiny_and_ragged_patches(oracle, gpu_ctx, frame, frame2, resx, resy, sm_kind):
    """N = 4 ... 1023 (below a wave, below a workgroup, not a multiple of 256): fused sums against the oracle's first
    iteration, materialised and lean."""
    corners = synth.square_corners(250, 244, 40)
    o_ssm = oracle.SSM(L.SSM_AFFINE, resx, resy); o_am = oracle.AM(L.AM_SSD, resx, resy); o_am.set_curr_img(frame)
    trk = oracle.Tracker(sm_kind, o_am, o_ssm, leven_marq=0, max_iters=1, epsilon=-1.0)
    trk.initialize(corners)
    o_am.set_curr_img(frame2)
    trk.update()
    rec = trk.trace()[0]
    for mat in (1, 0):
        gpu_ctx.set_image(frame)
        b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_AFFINE, resx, resy, 1)
        b.set_corners(corners[None])
        sm = mtf_amd.sm_desc(sm_kind, materialize=mat, leven_marq=0)
        b.init_template(sm)
        gpu_ctx.set_image(frame2)
        f, g, H = b.iterate(sm)
        assert abs(f[0] - rec["f"]) <= 1e-8 * abs(rec["f"]) + 1e-12
        assert rel(H[0], rec["H"]) < 1e-5
        gs = max(np.linalg.norm(rec["g"]), np.sqrt(abs(np.trace(rec["H"])) * abs(2 * rec["f"])))
        assert np.linalg.norm(g[0] - rec["g"]) <= 1e-5 * gs
        b.close()


def test_regions_outside_the_frame(oracle, gpu_ctx, frame):
    """A region wholly outside samples the constant border: It = 128 everywhere, zero gradient, H = 0 -- the device loop
    must leave such a target where it is while its neighbours in the same batch converge; a region half outside matches the
    oracle's border handling sample for sample."""
    h, w = frame.shape
    inside = synth.square_corners(256, 256, 80)
    outside = synth.square_corners(-300, -300, 80)
    half = synth.square_corners(w - 10, 200, 80)
    corners = np.stack([inside, outside, half])
    p_true = synth.random_small_homography(np.random.default_rng(4), 0.3)
    frame2 = synth.warp_frame(frame, p_true, (256.0, 256.0))
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 30, 30, 3)
    b.set_corners(corners)
    sm = mtf_amd.sm_desc(L.SM_FCLK, materialize=1, leven_marq=0, max_iters=20, epsilon=1e-5, hess_type=1)
    b.init_template(sm)
    assert np.all(b.read(L.BUF_I0)[1] == 128.0)
    o_ssm = oracle.SSM(L.SSM_HOMOGRAPHY, 30, 30); o_am = oracle.AM(L.AM_SSD, 30, 30); o_am.set_curr_img(frame)
    o_ssm.set_corners(half); o_am.initialize_pix_vals(o_ssm.get("curr_pts"))
    np.testing.assert_allclose(b.read(L.BUF_I0)[2], o_am.get("I0"), rtol=0, atol=1e-9)
    assert (o_am.get("I0") == 128.0).sum() > 300            # a good part of it is border
    gpu_ctx.set_image(frame2)
    f, g, H = b.iterate(sm)
    assert f[1] == 0.0 and np.all(g[1] == 0.0) and np.all(H[1] == 0.0)
    n_it, final = b.track(sm)
    np.testing.assert_allclose(final[1], outside, rtol=0, atol=0)        # singular system: no update, no NaN
    assert np.all(np.isfinite(final))
    W = synth.homography_from_state(p_true)
    q = W @ np.vstack([inside - 256.0, np.ones(4)])
    assert np.abs(final[0] - (q[:2] / q[2] + 256.0)).max() < 0.05       # the healthy neighbour converged


def test_degenerate_inputs(gpu_ctx, frame):
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 10, 10, 2)
    good = synth.square_corners(100, 100, 30)
    point = np.repeat(np.array([[50.0], [60.0]]), 4, axis=1)           # four coincident corners: no homography
    with pytest.raises(mtf_amd.InvalidArgument):
        b.set_corners(np.stack([good, point]))
    line = np.array([[10.0, 20.0, 30.0, 40.0], [10.0, 20.0, 30.0, 40.0]])   # collinear corners
    with pytest.raises(mtf_amd.InvalidArgument):
        b.set_corners(np.stack([good, line]))
    b.set_corners(np.stack([good, good + 40]))
    b.initialize_pix_vals(); b.initialize_similarity()
    one = b2 = None
    s = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 10, 10, 1)
    s.set_corners(good[None]); s.initialize_pix_vals(); s.initialize_similarity()
    lik = s.score_candidates(np.zeros((1, 8)))                          # a single candidate
    assert lik.shape == (1,) and abs(lik[0] - 1.0) < 1e-12    # (exactly 1 in MATH_REPLAY; the factored interpolant rounds differently)
    s.set_math_mode(mtf_amd.MATH_REPLAY)
    assert s.score_candidates(np.zeros((1, 8)))[0] == 1.0
    with pytest.raises(mtf_amd.InvalidArgument):
        s.score_candidates(np.zeros((0, 8)))
    with pytest.raises(mtf_amd.InvalidArgument):
        mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 10, 10, 0)
    with pytest.raises(mtf_amd.InvalidArgument):
        mtf_amd.Batch(gpu_ctx, L.AM_MI, L.SSM_HOMOGRAPHY, 10, 10, 1, mi_n_bins=17)


def test_flat_template_has_no_update(gpu_ctx):
    """A texture-less frame: J = 0, the constant Hessian is singular; every search method must return the region unchanged."""
    flat = np.full((256, 256), 77.0, dtype=np.float32)
    gpu_ctx.set_image(flat)
    c = synth.square_corners(128, 128, 50)
    for sm_kind in (L.SM_ESM, L.SM_FCLK, L.SM_ICLK):
        b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_AFFINE, 25, 25, 1)
        b.set_corners(c[None])
        sm = mtf_amd.sm_desc(sm_kind, materialize=0, leven_marq=0, max_iters=5, epsilon=1e-4)
        b.init_template(sm)
        # replay arithmetic: It has the bits of I0, the residual is exactly zero and so is the update.  Tolerance mode samples
        # with the factored interpolant: It differs from the (replay-initialised) template by ~1e-14, which the template's own
        # finite-difference noise (J0 ~ 1e-6 instead of 0 on a flat image) turns into an update of ~1e-8 px -- noise over noise
        for mode, atol in ((mtf_amd.MATH_REPLAY, 0.0), (mtf_amd.MATH_FAST, 1e-6)):
            b.set_math_mode(mode)
            b.set_corners(c[None])
            n_it, final = b.track(sm)
            np.testing.assert_allclose(final[0], c, rtol=0, atol=atol)
            assert n_it[0] == 1                                          # converged at once: zero update
        b.close()


@pytest.mark.gpu
def test_row_pair_image_gives_the_same_scores_and_follows_the_image(gpu_ctx, monkeypatch):
    """r06: the candidate scorer gathers a bilinear cell from the row-pair copy of an UPLOADED frame (one 16-byte load: pair[2 (y W + x)] =
    I[y][x] | I[y + 1][x]) -- the same four texels as the two 8-byte gathers, so the same bits (MTFHIP_PAIR_IMAGE=0: the two gathers); the
    copy is rebuilt when the image is replaced (upload, keep_prev + upload, swap_prev) and not used for a borrowed image.  (It is built once
    enough candidates have been scored on an image, MTFHIP_PAIR_IMAGE_AFTER: 0 here.)"""
    monkeypatch.setenv("MTFHIP_PAIR_IMAGE_AFTER", "0")
    import torch
    import mtf_amd
    from mtf_amd import _lib as L
    from mtf_amd import synth
    f0, f1 = synth.make_frame(384, 416), synth.make_frame(384, 416, seed=77)
    corners = synth.square_corners(200, 180, 90)
    rng = np.random.default_rng(3)
    states = rng.normal(size=(513, 8)) * np.array([0.02, 0.02, 3.0, 0.02, 0.02, 3.0, 1e-4, 1e-4])
    states[:40, 2] -= 160.0          # some candidates across the left border: the per-sample path

    def scores(img, pair, borrow=None):
        monkeypatch.setenv("MTFHIP_PAIR_IMAGE", pair)
        if borrow is None:
            gpu_ctx.set_image(img)
        else:
            gpu_ctx.set_image_device(borrow.data_ptr(), img.shape[0], img.shape[1], keep=borrow)
        b = mtf_amd.Batch(gpu_ctx, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, 40, 1)
        b.set_corners(corners[None]); b.initialize_pix_vals(); b.initialize_similarity()
        lik, sim = b.score_candidates(states, want_similarity=True)
        _, rows = b.nn_dataset(64, np.array([0.01, 0.01, 2.0, 0.01, 0.01, 2.0, 1e-5, 1e-5]), None, seed=4)
        b.close()
        return lik, sim, rows
    a = scores(f0, "1"); b0 = scores(f0, "0")
    for x, y in zip(a, b0):
        assert np.array_equal(x, y)
    c = scores(f1, "1"); d = scores(f1, "0")          # a new frame: the copy follows it
    for x, y in zip(c, d):
        assert np.array_equal(x, y)
    assert not np.array_equal(a[1], c[1])
    gpu_ctx.keep_prev(); e = scores(f0, "1")           # the other buffer of the context
    for x, y in zip(e, b0):
        assert np.array_equal(x, y)
    gpu_ctx.swap_prev()                                 # current <-> previous: f1 again (swap back before the next upload)
    monkeypatch.setenv("MTFHIP_PAIR_IMAGE", "1")
    b = mtf_amd.Batch(gpu_ctx, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, 40, 1)
    b.set_corners(corners[None]); b.initialize_pix_vals(); b.initialize_similarity()
    assert np.array_equal(b.score_candidates(states, want_similarity=True)[1], d[1])
    b.close()
    gpu_ctx.swap_prev()
    t = torch.from_numpy(f1).to("cuda:0")
    g = scores(f1, "1", borrow=t)                       # a borrowed image: no copy, the two gathers
    for x, y in zip(g, d):
        assert np.array_equal(x, y)
