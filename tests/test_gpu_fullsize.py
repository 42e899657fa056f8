"""GPU: BASELINE.json's configurations at their FULL sizes, checked through size-independent properties
(the oracle comparisons at sizes it finishes in seconds are in test_gpu_parity.py / test_gpu_trackers.py):
zero residual at the identity, the fused sums against a float64 checksum of the materialised arrays, batch
independence (a target inside a 64-wide batch gives the bits it gives alone), permutation / sharding
equivariance of candidate scoring, convergence to a known synthetic warp."""
import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import synth
from mtf_amd.sm import GridTracker, LKTracker, NTSearchMethod

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_frames():
    f0 = synth.make_frame(1024, 1024)
    p_true = synth.random_small_homography(np.random.default_rng(77), 0.3)
    return f0, synth.warp_frame(f0, p_true, (512.0, 512.0)), p_true


def gt_corners(corners, p_true, centre=(512.0, 512.0)):
    W = synth.homography_from_state(p_true)
    c = np.asarray(centre)[:, None]
    q = W @ np.vstack([corners - c, np.ones(corners.shape[1])])
    return q[:2] / q[2] + c


@pytest.mark.parametrize("chained", [1, 0])
def test_config2_fclk_200x200_checksums(gpu_ctx, big_frames, chained):
    """Config 2: FCLK + SSD + Homography, 200 x 200, both chained_warp settings."""
    f0, f1, p_true = big_frames
    corners = synth.square_corners(512, 512, 200)
    gpu_ctx.set_image(f0)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, 1)
    b.set_corners(corners[None])
    sm = mtf_amd.sm_desc(L.SM_FCLK, chained_warp=chained, hess_type=1, materialize=1)
    b.init_template(sm)
    # same frame, identity warp: the residual is exactly zero, so are f and g; H is the negated Gram of J0
    f, g, H = b.iterate(sm)
    assert f[0] == 0.0 and np.all(g[0] == 0.0)
    J = b.read(L.BUF_JT)[0]
    np.testing.assert_array_equal(J, b.read(L.BUF_J0)[0])
    np.testing.assert_allclose(H[0], -(J.T @ J), rtol=1e-12)
    assert np.linalg.eigvalsh(H[0]).max() < 0
    # next frame: the fused sums against float64 checksums of what the same launch materialised
    gpu_ctx.set_image(f1)
    f, g, H = b.iterate(sm)
    It, I0, J = b.read(L.BUF_IT)[0], b.read(L.BUF_I0)[0], b.read(L.BUF_JT)[0]
    r = It - I0
    assert abs(f[0] + 0.5 * (r @ r)) <= 1e-12 * abs(f[0])
    np.testing.assert_allclose(H[0], -(J.T @ J), rtol=1e-12)
    np.testing.assert_allclose(g[0], -(r @ J), rtol=1e-9, atol=1e-12 * np.abs(J).max() * np.abs(r).sum())
    # lean mode (nothing materialised) produces the same sums bit for bit
    lean = mtf_amd.sm_desc(L.SM_FCLK, chained_warp=chained, hess_type=1, materialize=0)
    b.set_math_mode(mtf_amd.MATH_REPLAY)
    f2, g2, H2 = b.iterate(lean)
    assert f2[0] == f[0] and np.array_equal(g2, g) and np.array_equal(H2, H)
    # ... and in tolerance mode (FMA, one reciprocal per point, closed-form gradient) within the north-star budget, plain relative
    b.set_math_mode(mtf_amd.MATH_FAST)
    f3, g3, H3 = b.iterate(lean)
    assert abs(f3[0] - f[0]) <= 1e-9 * abs(f[0])
    assert np.linalg.norm(H3 - H) <= 1e-5 * np.linalg.norm(H)
    assert np.linalg.norm(g3 - g) <= 1e-5 * np.linalg.norm(g)
    dp, dp3 = np.linalg.solve(H[0], -g[0]), np.linalg.solve(H3[0], -g3[0])
    assert np.linalg.norm(dp3 - dp) <= 1e-5 * np.linalg.norm(dp)
    # and the loop converges on the known warp
    trk = LKTracker(gpu_ctx, L.SM_FCLK, L.SSM_HOMOGRAPHY, 200, 200, 1, host_solve=False, chained_warp=chained,
                    hess_type=1, max_iters=30, epsilon=1e-6)
    gpu_ctx.set_image(f0); trk.initialize(corners[None]); gpu_ctx.set_image(f1)
    out = trk.update()[0]
    assert np.abs(out - gt_corners(corners, p_true)).max() < 0.05


def test_headline_esm_200x200_batch_independence(gpu_ctx, big_frames):
    """The metric's workload (ESM + SSD + Homography, 200 x 200, 64 targets per launch): every target of the batch
    produces exactly the bits it produces alone, and lands on the ground-truth region."""
    f0, f1, p_true = big_frames
    B = 64
    rng = np.random.default_rng(5)
    cx = rng.uniform(250, 774, B); cy = rng.uniform(250, 774, B)
    corners = np.stack([synth.square_corners(cx[i], cy[i], 200) for i in range(B)])
    sm = mtf_amd.sm_desc(L.SM_ESM, materialize=1)
    gpu_ctx.set_image(f0)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, B)
    b.set_corners(corners); b.init_template(sm)
    gpu_ctx.set_image(f1)
    f, g, H = b.iterate(sm)
    for t in (0, 17, 63):
        gpu_ctx.set_image(f0)
        s = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, 1)
        s.set_corners(corners[t][None]); s.init_template(sm)
        gpu_ctx.set_image(f1)
        fs, gs, Hs = s.iterate(sm)
        # per-pixel results are identical; the work decomposition (hence the summation tree) depends on B
        np.testing.assert_array_equal(s.read(L.BUF_JT)[0], b.read(L.BUF_JT)[t])
        np.testing.assert_allclose(Hs[0], H[t], rtol=1e-12)
        np.testing.assert_allclose(gs[0], g[t], rtol=1e-9, atol=1e-9 * np.abs(g[t]).max())
        assert abs(fs[0] - f[t]) <= 1e-12 * abs(f[t])
        s.close()
    trk_sm = mtf_amd.sm_desc(L.SM_ESM, materialize=0, max_iters=30, epsilon=1e-6)
    gpu_ctx.set_image(f0); b.set_corners(corners); b.init_template(trk_sm); gpu_ctx.set_image(f1)
    n_it, out = b.track(trk_sm)
    for t in range(B):
        assert np.abs(out[t] - gt_corners(corners[t], p_true)).max() < 0.05, t
    assert n_it.max() < 30


def test_config3_grid_256_patches(gpu_ctx, big_frames):
    """Config 3: 16 x 16 patches, ICLK + NCC + Affine 25 x 25, the whole patch loop in one launch."""
    f0, f1, p_true = big_frames
    region = synth.square_corners(512, 512, 400)
    gpu_ctx.set_image(f0)
    gt = GridTracker(gpu_ctx, grid_size=16, patch_size=25, am=L.AM_NCC, ssm=L.SSM_AFFINE, max_iters=30, epsilon=1e-4)
    gt.initialize(region)
    patches = gt.patch_corners(region)
    assert patches.shape == (256, 2, 4)
    gpu_ctx.set_image(f1)
    corners, centroids = gt.update_patches()
    want = np.stack([gt_corners(patches[k], p_true).mean(axis=1) for k in range(256)])
    err = np.abs(centroids - want).max(axis=1)
    assert np.median(err) < 0.05 and (err < 0.5).mean() > 0.97
    # idempotence: re-initialised on the same frame, a second update leaves every patch where it is
    gpu_ctx.set_image(f0); gt.initialize(region)
    c0, _ = gt.update_patches()
    np.testing.assert_allclose(c0, patches, atol=1e-6)
    # the frame loop as GridTracker::update runs it -- every patch tracker reset to the grid, then updated -- in ONE call
    # (mtfhip_batch_track_region) against the two calls, in both arithmetic modes
    for math in (mtf_amd.MATH_FAST, mtf_amd.MATH_REPLAY):
        gt.tracker.batch.set_math_mode(math)
        gpu_ctx.set_image(f1)
        gt.tracker.set_region(patches); two, _ = gt.update_patches()
        one, cen = gt.update_patches(region)
        assert np.array_equal(one, two)
        err = np.abs(cen - want).max(axis=1)
        assert np.median(err) < 0.05 and (err < 0.5).mean() > 0.97


@pytest.mark.parametrize("am,ssm,patch", [(L.AM_NCC, L.SSM_AFFINE, 25), (L.AM_SSD, L.SSM_HOMOGRAPHY, 30)])
def test_grid_frame_region_mode_equals_separate_reset(gpu_ctx, big_frames, am, ssm, patch):
    """r04: the grid frame is ONE launch -- every workgroup of k_iclk_track ingests its patch's region corners from the pinned staging
    buffer, derives the square-to-quadrilateral map and lays out its own grid (RegionIngest) -- against the r03 form (slab ingest +
    k_init_grid in a launch of their own, MTFHIP_GRID_FUSED=0): bit-identical corners, iteration counts AND template grids (the
    kernel also writes INIT_PTS / INIT_HXY / INIT_Z for the calls that come after the frame), in both arithmetic modes; a patch with
    degenerate corners is reported as the separate reset reports it."""
    import os
    f0, f1, p_true = big_frames
    region = synth.square_corners(512, 512, 400)
    if ssm == L.SSM_HOMOGRAPHY:   # a general quadrilateral: the patch grids are projective (unit_z = 0)
        region = region + np.array([[3.0, -2.0, 5.0, -4.0], [-1.5, 2.5, 4.0, -3.0]])
    gpu_ctx.set_image(f0)
    gt = GridTracker(gpu_ctx, grid_size=8, patch_size=patch, am=am, ssm=ssm, max_iters=10, epsilon=1e-4)
    gt.initialize(region)
    b = gt.tracker.batch
    gpu_ctx.set_image(f1)
    out = {}
    for math in (mtf_amd.MATH_FAST, mtf_amd.MATH_REPLAY):
        b.set_math_mode(math)
        for mode in ("0", "1"):
            os.environ["MTFHIP_GRID_FUSED"] = mode
            try:
                c, cen = gt.update_patches(region)
                out[mode] = (c.copy(), cen.copy(), gt.tracker.n_iters.copy() if hasattr(gt.tracker, "n_iters") else None,
                             b.read(L.BUF_INIT_PTS).copy(), b.read(L.BUF_INIT_HXY).copy(), b.read(L.BUF_INIT_Z).copy(), b.get_state().copy())
            finally:
                del os.environ["MTFHIP_GRID_FUSED"]
        for x, y, what in zip(out["0"], out["1"], ("corners", "centroids", "n_iters", "init_pts", "init_hxy", "init_z", "state")):
            if x is not None:
                assert np.array_equal(x, y), what
    # degenerate corners (three collinear points) in one patch: an error either way
    bad = gt.patch_corners(region).copy()
    bad[5] = np.array([[100.0, 110.0, 120.0, 130.0], [200.0, 200.0, 200.0, 200.0]])
    for mode in ("0", "1"):
        os.environ["MTFHIP_GRID_FUSED"] = mode
        try:
            with pytest.raises(mtf_amd.MtfHipError, match="degenerate"):
                gt.update_patches(bad)
        finally:
            del os.environ["MTFHIP_GRID_FUSED"]
    gt.update_patches(region)   # and the tracker is usable afterwards


def test_config4_pf_10000_candidates(gpu_ctx, big_frames):
    """Config 4: 10 000 candidates x 2 500 px: permutation and sharding equivariance, identity candidate."""
    f0, f1, _ = big_frames
    rng = np.random.default_rng(9)
    gpu_ctx.set_image(f0)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 50, 50, 1)
    b.set_corners(synth.square_corners(512, 512, 100)[None]); b.initialize_pix_vals(); b.initialize_similarity()
    states = synth.pf_candidate_states(rng, 10000)
    states[1234] = 0
    b.set_math_mode(mtf_amd.MATH_REPLAY)      # the reference's arithmetic: the identity candidate reproduces the template bit for bit
    lik, sim = b.score_candidates(states, want_similarity=True)
    assert lik.shape == (10000,) and sim[1234] == 0.0 and lik[1234] == 1.0 and np.all(sim <= 0) and np.all(lik <= 1)
    b.set_math_mode(mtf_amd.MATH_FAST)        # tolerance mode (the default): the same scores to 1e-9, the identity to rounding
    lik_r, sim_r = lik, sim
    lik, sim = b.score_candidates(states, want_similarity=True)
    np.testing.assert_allclose(lik, lik_r, rtol=1e-9, atol=1e-300)
    assert abs(sim[1234]) < 1e-18 and abs(lik[1234] - 1.0) < 1e-10 and np.all(sim <= 0) and np.all(lik <= 1)
    perm = rng.permutation(10000)
    lik_p = b.score_candidates(states[perm])
    assert np.array_equal(lik_p, lik[perm])                                   # a candidate's score does not depend on its slot
    shards = np.concatenate([b.score_candidates(states[k * 1250:(k + 1) * 1250]) for k in range(8)])
    assert np.array_equal(shards, lik)                                        # an 8-way partition of the candidates gathers to the same vector
    gpu_ctx.set_image(f1)
    assert b.score_candidates(states[:100]).max() < 1.0


@pytest.mark.parametrize("exchange", ["collective", "peer"])
def test_config4_pf_sharded_world8_full_size(big_frames, exchange):
    """Config 4 as `bench.py --workload pf --gpus 8` runs it, at its full size: 10 000 particles x 2 500 px sharded over EIGHT ranks
    (threads of this process over the loopback communicator: block bounds, one in-place all-gather of 1 250 weights per rank,
    replicated proposals / scan / selection) -- every rank ends four iterations with the unsharded filter's particle set, weights
    and estimate, bit for bit, and the estimate follows the known synthetic warp.  exchange="peer": the weights travel as stores
    of the scoring kernel into the eight mailboxes instead (mtfhip_pf_set_exchange), same bits."""
    from mtf_amd.sm import Comm, ParticleFilter
    from test_gpu_trackers import _run_ranks
    f0, f1, p_true = big_frames
    corners = synth.square_corners(512, 512, 100)
    kw = dict(n_particles=10000, ssm_sigma=(1.0, 0.5, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, likelihood_alpha=5.0, seed=11, mean_type=1)

    def run(comm):
        ctx = mtf_amd.Context(0)
        ctx.set_image(f0)
        pf = ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 50, 50, comm=comm, exchange=exchange if comm is not None else "collective", **kw)
        pf.initialize(corners[None])
        ctx.set_image(f1)
        for _ in range(4):
            pf.iteration()
        st, ar, w, ids = pf.particles()
        out = (st.copy(), w.copy(), ids.copy(), pf.get_region().copy())
        pf.close(); ctx.close()
        return out
    ref = run(None)
    comms = Comm.loopback(8)
    got = _run_ranks(8, lambda r: run(comms[r]))
    for c in comms:
        c.close()
    for r in range(8):
        for a, b, what in zip(got[r], ref, ("states", "weights", "ids", "corners")):
            assert np.array_equal(a, b), "rank %d: %s differ from the unsharded filter" % (r, what)
    err = np.linalg.norm(ref[3][0] - gt_corners(corners, p_true), axis=0)
    assert err.max() < 3.0     # a particle filter's estimate: within a few pixels of the true warp after four iterations


def test_config5_mi_400x400_64_targets(gpu_ctx):
    """Config 5: ESM + MI (8 bins) + Homography, 400 x 400, 64 concurrent targets on a 2048 x 2048 frame."""
    f0 = synth.make_frame(2048, 2048)
    p_true = synth.random_small_homography(np.random.default_rng(3), 0.2)
    f1 = synth.warp_frame(f0, p_true, (1024.0, 1024.0))
    B = 64
    rng = np.random.default_rng(11)
    cx = rng.uniform(450, 1600, B); cy = rng.uniform(450, 1600, B)
    corners = np.stack([synth.square_corners(cx[i], cy[i], 400) for i in range(B)])
    gpu_ctx.set_image(f0)
    nt = NTSearchMethod(gpu_ctx, L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 400, 400, B, max_iters=4, epsilon=-1.0)
    nt.initialize(corners)
    H0 = nt.H0.copy()
    for t in range(0, B, 9):
        assert np.allclose(H0[t], H0[t].T, rtol=1e-9, atol=1e-9 * np.abs(H0[t]).max())
        assert np.linalg.eigvalsh(0.5 * (H0[t] + H0[t].T)).max() < 0                 # MI is at its maximum on the template
    gpu_ctx.set_image(f1)
    nt.update()
    f_first, f_last = nt.trace[0]["f"], nt.batch.get_similarity()
    assert np.all(f_last > f_first)                                                   # every target's MI rises
    want = np.stack([gt_corners(corners[t], p_true, (1024.0, 1024.0)) for t in range(B)])
    before = np.abs(corners - want).max(axis=(1, 2)); after = np.abs(nt.get_region() - want).max(axis=(1, 2))
    assert np.all(after < 0.5 * before)
    # batch independence at full size: target 20 alone reproduces its first-iteration g and H
    gpu_ctx.set_image(f0)
    one = NTSearchMethod(gpu_ctx, L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 400, 400, 1, max_iters=1, epsilon=-1.0)
    one.initialize(corners[20][None])
    gpu_ctx.set_image(f1)
    one.update()
    np.testing.assert_allclose(one.trace[0]["H"][0], nt.trace[0]["H"][20], rtol=1e-10)
    np.testing.assert_allclose(one.trace[0]["g"][0], nt.trace[0]["g"][20], rtol=1e-8, atol=1e-12)
    assert abs(one.trace[0]["f"][0] - nt.trace[0]["f"][20]) <= 1e-12 * abs(nt.trace[0]["f"][20])


@pytest.mark.parametrize("math", ["fast", "replay"])
def test_config5_mi_device_loop_full_size(gpu_ctx, math):
    """Config 5 at full size through mtfhip_batch_track -- the recompute passes (k_mi_pass_hist, k_mi_pass_grad_hess: nothing N-sized
    written) + the device-side finish: the trajectory of the call-by-call search method above (the interface-level path
    over the materialising kernels) is followed by every target, MI rises, and the regions land where the ground truth says."""
    from mtf_amd.sm import LKTracker
    f0 = synth.make_frame(2048, 2048)
    p_true = synth.random_small_homography(np.random.default_rng(3), 0.2)
    f1 = synth.warp_frame(f0, p_true, (1024.0, 1024.0))
    B = 64
    rng = np.random.default_rng(11)
    cx = rng.uniform(450, 1600, B); cy = rng.uniform(450, 1600, B)
    corners = np.stack([synth.square_corners(cx[i], cy[i], 400) for i in range(B)])
    n_it = 4
    gpu_ctx.set_image(f0)
    nt = NTSearchMethod(gpu_ctx, L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 400, 400, B, max_iters=n_it, epsilon=-1.0)
    nt.initialize(corners)
    gpu_ctx.set_image(f1)
    nt.update()
    ref = nt.get_region().copy()
    nt.batch.close()
    gpu_ctx.set_image(f0)
    trk = LKTracker(gpu_ctx, L.SM_ESM, L.SSM_HOMOGRAPHY, 400, 400, B, host_solve=False, am=L.AM_MI, max_iters=n_it, epsilon=-1.0,
                    leven_marq=0, materialize=0)
    trk.batch.set_math_mode(mtf_amd.MATH_FAST if math == "fast" else mtf_amd.MATH_REPLAY)
    trk.initialize(corners)
    gpu_ctx.set_image(f1)
    out = trk.update()
    assert np.all(trk.n_iters == n_it)
    want = np.stack([gt_corners(corners[t], p_true, (1024.0, 1024.0)) for t in range(B)])
    before = np.abs(corners - want).max(axis=(1, 2)); after = np.abs(out - want).max(axis=(1, 2))
    assert np.all(after < 0.5 * before)
    # the same trajectory as the call-by-call path: MI's update moves by 1.5-2.8e-5 relative under a one-ulp change of the grid
    # (test_mi_update_noise_floor), four iterations of 400-pixel-wide corners stay within a few 1e-3 px of each other
    assert np.abs(out - ref).max() < 5e-3
    # replay arithmetic: a target tracked alone follows the same path (batch independence of the recompute passes)
    if math == "replay":
        gpu_ctx.set_image(f0)
        one = LKTracker(gpu_ctx, L.SM_ESM, L.SSM_HOMOGRAPHY, 400, 400, 1, host_solve=False, am=L.AM_MI, max_iters=n_it, epsilon=-1.0,
                        leven_marq=0, materialize=0)
        one.batch.set_math_mode(mtf_amd.MATH_REPLAY)
        one.initialize(corners[20][None])
        gpu_ctx.set_image(f1)
        np.testing.assert_allclose(one.update()[0], out[20], rtol=0, atol=1e-6)   # (its own workgroup decomposition: summation order only)


def test_multichannel_200x200x3_fused_iteration(gpu_ctx):
    """MCSSD at the headline patch size (200 x 200 x 3 rows per target) through the fused iteration: a target's g / H do not depend
    on the batch it is in, replicated channels give three times the single-channel sums, and the device loop converges."""
    f0 = synth.make_frame_mc(1024, 1024)
    p_true = synth.random_small_homography(np.random.default_rng(5), 0.3)
    f1 = synth.warp_frame(f0, p_true, (512.0, 512.0))
    B = 6
    rng = np.random.default_rng(9)
    corners = np.stack([synth.square_corners(rng.uniform(300, 700), rng.uniform(300, 700), 200.0) for _ in range(B)])
    sm = mtf_amd.sm_desc(L.SM_ESM, materialize=1, leven_marq=0, max_iters=10, epsilon=1e-6)
    gpu_ctx.set_image(f0)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, B, n_channels=3)
    b.set_corners(corners); b.init_template(sm)
    gpu_ctx.set_image(f1)
    f, g, H = b.iterate(sm)
    gpu_ctx.set_image(f0)
    one = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, 1, n_channels=3)
    one.set_corners(corners[4][None]); one.init_template(sm)
    gpu_ctx.set_image(f1)
    f1_, g1, H1 = one.iterate(sm)
    np.testing.assert_allclose(H1[0], H[4], rtol=1e-11)
    np.testing.assert_allclose(g1[0], g[4], rtol=1e-9, atol=1e-9 * np.abs(g[4]).max())
    assert abs(f1_[0] - f[4]) <= 1e-12 * abs(f[4])
    # three copies of one channel: every row sum is three times the single-channel one
    g0 = np.repeat(f0[..., :1], 3, axis=2).copy(); g1f = np.repeat(f1[..., :1], 3, axis=2).copy()
    gpu_ctx.set_image(g0)
    rep = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, 1, n_channels=3)
    rep.set_corners(corners[1][None]); rep.init_template(sm)
    gpu_ctx.set_image(g1f)
    fr, gr, Hr = rep.iterate(sm)
    gpu_ctx.set_image(np.ascontiguousarray(f0[..., 0]))
    sc = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 200, 1)
    sc.set_corners(corners[1][None]); sc.init_template(sm)
    gpu_ctx.set_image(np.ascontiguousarray(f1[..., 0]))
    fs, gs, Hs = sc.iterate(sm)
    np.testing.assert_allclose(Hr[0], 3 * Hs[0], rtol=1e-6)      # (mc:: forms the bilinear weights first: rounding differs from the single-channel sampler)
    np.testing.assert_allclose(gr[0], 3 * gs[0], rtol=1e-6, atol=1e-6 * np.abs(gs[0]).max())
    assert abs(fr[0] - 3 * fs[0]) <= 1e-9 * abs(fs[0])
    # device loop
    gpu_ctx.set_image(f1)
    n_it, out = b.track(sm)
    want = np.stack([gt_corners(corners[t], p_true, (512.0, 512.0)) for t in range(B)])
    assert np.abs(out - want).max() < 0.1 and n_it.max() <= 10
    for x in (b, one, rep, sc):
        x.close()
