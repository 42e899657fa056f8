"""GPU: the search-method drivers (mtf_amd/sm.py) end to end -- the callers of the hot path."""
import os
import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import synth
from mtf_amd.sm import GridTracker, LKTracker, NTSearchMethod, ParticleFilter

pytestmark = pytest.mark.gpu


def gt_corners(corners, p_true, centre):
    W = synth.homography_from_state(p_true)
    q = W @ np.vstack([corners - np.array(centre)[:, None], np.ones(4)])
    return q[:2] / q[2] + np.array(centre)[:, None]


@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC, L.AM_MI])
@pytest.mark.parametrize("host_solve", [True, False])
@pytest.mark.parametrize("sm", [L.SM_ESM, L.SM_FCLK, L.SM_ICLK])
def test_lk_trackers_recover_known_warp(gpu_ctx, frame, sm, host_solve, am):
    if am == L.AM_MI and sm != L.SM_ESM:
        # MI with 8 bins on a 45 x 45 patch is a flat objective: FCLK / ICLK stall 0.8-1 px from the truth on the device exactly
        # as on the host (test_device_side_loop_mi and test_fused_mi_iterations_follow_oracle hold them to the oracle)
        pytest.skip("MI + FCLK / ICLK do not reach 0.08 px on this patch (neither does the oracle)")
    rng = np.random.default_rng(7)
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], 90)
    p_true = synth.random_small_homography(rng, 0.4)
    frame2 = synth.warp_frame(frame, p_true, centre)
    gpu_ctx.set_image(frame)
    trk = LKTracker(gpu_ctx, sm, L.SSM_HOMOGRAPHY, 45, 45, 1, host_solve=host_solve, max_iters=40, epsilon=1e-6,
                    materialize=0, am=am)
    trk.initialize(corners[None])
    gpu_ctx.set_image(frame2)
    out = trk.update()
    assert np.abs(out[0] - gt_corners(corners, p_true, centre)).max() < 0.08
    assert 2 <= int(trk.n_iters[0]) <= 40


@pytest.mark.parametrize("host_solve", [True, False])
def test_lm_host_loop_matches_oracle(oracle, gpu_ctx, frame, host_solve):
    """Levenberg-Marquardt accept / undo -- on the host with device f, g, H, and inside the device-side loop
    (mtfhip_batch_track: the test and the undo are part of k_finish_track): same corners as the oracle's nt:: trackers with
    leven_marq = 1 (the class default, ESMParams.cc:4-15) on a motion large enough to trigger rejections."""
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], 100)
    p_true = np.array([0.01, -0.01, 6.0, 0.01, 0.0, -5.0, 0, 0])
    frame2 = synth.warp_frame(frame, p_true, centre)
    for sm, am in ((L.SM_FCLK, L.AM_SSD), (L.SM_ESM, L.AM_SSD), (L.SM_ICLK, L.AM_SSD), (L.SM_ESM, L.AM_NCC), (L.SM_FCLK, L.AM_NCC)):
        o_ssm = oracle.SSM(0, 40, 40); o_am = oracle.AM(am, 40, 40); o_am.set_curr_img(frame)
        otrk = oracle.Tracker(sm, o_am, o_ssm, leven_marq=1, max_iters=60, epsilon=1e-8)
        otrk.initialize(corners)
        o_am.set_curr_img(frame2)
        o_iters = otrk.update()
        gpu_ctx.set_image(frame)
        trk = LKTracker(gpu_ctx, sm, L.SSM_HOMOGRAPHY, 40, 40, 1, host_solve=host_solve, leven_marq=1, max_iters=60,
                        epsilon=1e-8, materialize=0, am=am)
        trk.initialize(corners[None])
        gpu_ctx.set_image(frame2)
        out = trk.update()
        np.testing.assert_allclose(out[0], otrk.get_region(), atol=5e-4)
        if not host_solve:   # the device loop counts passes like the reference's iters_done
            assert abs(int(trk.n_iters[0]) - o_iters) <= 2, (sm, am, trk.n_iters, o_iters)
        assert np.abs(out[0] - gt_corners(corners, p_true, centre)).max() < 0.1


@pytest.mark.parametrize("am,alpha", [(L.AM_SSD, 5.0), (L.AM_NCC, 2000.0)])
def test_particle_filter_follows_translation(gpu_ctx, frame, am, alpha):
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], 80)
    p_true = np.array([0, 0, 3.0, 0, 0, -2.0, 0, 0])
    frame2 = synth.warp_frame(frame, p_true, centre)
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 40, 40, n_particles=4000, max_iters=4, epsilon=1e-9, seed=3,
                        ssm_sigma=(0.002, 0.002, 1.5, 0.002, 0.002, 1.5, 1e-6, 1e-6), likelihood_alpha=alpha, am=am)
    pf.initialize(corners[None])
    gpu_ctx.set_image(frame2)
    out = pf.update()
    assert np.abs(out[0] - gt_corners(corners, p_true, centre)).max() < 1.0


NT_CASES = [
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, 25, dict()),                       # config 3 patch tracker
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, 25, dict(hess_type=2)),
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, 25, dict(hess_type=1, chained_warp=0)),
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, dict()),
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, dict(jac_type=0, hess_type=3)),
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, dict(hess_type=4)),
    (L.SM_FCLK, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, dict(hess_type=2)),
    (L.SM_FCLK, L.AM_NCC, L.SSM_AFFINE, 30, dict()),
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 40, dict()),
    (L.SM_FCLK, L.AM_SSD, L.SSM_AFFINE, 30, dict(chained_warp=0)),
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict()),                      # config 5 (reduced)
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict(jac_type=0, hess_type=3)),
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict(hess_type=4)),
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict(hess_type=5, chained_warp=0)),
    (L.SM_FCLK, L.AM_MI, L.SSM_AFFINE, 30, dict()),
    (L.SM_ICLK, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict()),
    (L.SM_ICLK, L.AM_MI, L.SSM_AFFINE, 30, dict(hess_type=2)),
    # second-order Hessians (sec_ord_hess = 1), every branch of NT/ESM.cc:315-377, NT/FCLK.cc:262-283, NT/ICLK.cc:204-252
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 40, dict(sec_ord_hess=1, hess_type=5)),
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 40, dict(sec_ord_hess=1, hess_type=4, chained_warp=0)),
    (L.SM_ESM, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=3, jac_type=0)),
    (L.SM_ESM, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1)),
    (L.SM_FCLK, L.AM_SSD, L.SSM_HOMOGRAPHY, 40, dict(sec_ord_hess=1, hess_type=2)),
    (L.SM_ICLK, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=2)),
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, 25, dict(sec_ord_hess=1, hess_type=2)),
    (L.SM_FCLK, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, dict(sec_ord_hess=1, hess_type=2, chained_warp=0)),
    (L.SM_ESM, L.AM_NCC, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=4)),
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict(sec_ord_hess=1)),
    (L.SM_ICLK, L.AM_MI, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=1)),
    (L.SM_FCLK, L.AM_MI, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=0)),
]


@pytest.mark.parametrize("n_bins", [5, 12, 16])
@pytest.mark.parametrize("sm_kind,extra", [(L.SM_ESM, dict()), (L.SM_ESM, dict(hess_type=4, jac_type=0)), (L.SM_FCLK, dict(hess_type=2)),
                                           (L.SM_ICLK, dict()), (L.SM_ESM, dict(sec_ord_hess=1))],
                         ids=["esm", "esm_sumofstd_original", "fclk_std", "iclk", "esm_second_order"])
def test_mi_histogram_sizes(oracle, gpu_ctx, frame, n_bins, sm_kind, extra):
    """MI with other histogram sizes than the reference's default 8 (mi_n_bins of the patch descriptor; MI.cc:14-15,97-104):
    the VALU bin mode (the FP64 MFMA tiles cover 8 bins), 16 bins = the widest rows, an odd count = ragged bin pairs."""
    rng = np.random.default_rng(31)
    res, centre = 36, (250.0, 262.0)
    corners = synth.square_corners(centre[0], centre[1], 60.0)
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.3), centre)
    params = dict(leven_marq=0, max_iters=3, epsilon=-1.0)
    params.update(extra)
    o_ssm = oracle.SSM(L.SSM_HOMOGRAPHY, res, res); o_am = oracle.AM(L.AM_MI, res, res, n_bins=n_bins); o_am.set_curr_img(frame)
    otrk = oracle.Tracker(sm_kind, o_am, o_ssm, **params)
    otrk.initialize(corners)
    gpu_ctx.set_image(frame)
    nt = NTSearchMethod(gpu_ctx, sm_kind, L.AM_MI, L.SSM_HOMOGRAPHY, res, res, 1, am_params=dict(mi_n_bins=n_bins), **params)
    nt.initialize(corners[None])
    o_am.set_curr_img(frame2); gpu_ctx.set_image(frame2)
    otrk.update(); nt.update()
    rec, got = otrk.trace()[0], nt.trace[0]
    assert abs(got["f"][0] - rec["f"]) <= 1e-7 * abs(rec["f"])
    assert np.linalg.norm(got["H"][0] - rec["H"]) <= 1e-5 * np.linalg.norm(rec["H"])
    gs = max(np.linalg.norm(rec["g"]), 1e-3 * np.sqrt(abs(np.trace(rec["H"]))))
    assert np.linalg.norm(got["g"][0] - rec["g"]) <= 1e-4 * gs
    if not extra.get("sec_ord_hess"):
        np.testing.assert_allclose(nt.get_region()[0], otrk.get_region(), atol=2e-4)
    if not extra:
        # the fused MI iteration (mtfhip_batch_iterate) at the same histogram size
        gpu_ctx.set_image(frame)
        b = mtf_amd.Batch(gpu_ctx, L.AM_MI, L.SSM_HOMOGRAPHY, res, res, 1, mi_n_bins=n_bins)
        b.set_corners(corners[None])
        sm = mtf_amd.sm_desc(sm_kind, **params)
        b.init_template(sm)
        gpu_ctx.set_image(frame2)
        f, g, H = b.iterate(sm)
        assert abs(f[0] - rec["f"]) <= 1e-7 * abs(rec["f"])
        assert np.linalg.norm(H[0] - rec["H"]) <= 1e-5 * np.linalg.norm(rec["H"])
        assert np.linalg.norm(g[0] - rec["g"]) <= 1e-4 * gs
        b.close()


def oracle_first_order_H(oracle, case, frame, frame2, corners):
    """H of the first iteration with sec_ord_hess = 0 (to show that the second-order term is not vacuous)"""
    sm_kind, am, ssm, res, extra = case
    params = dict(leven_marq=0, max_iters=1, epsilon=-1.0)
    params.update(extra); params["sec_ord_hess"] = 0
    o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(am, res, res); o_am.set_curr_img(frame)
    trk = oracle.Tracker(sm_kind, o_am, o_ssm, **params)
    trk.initialize(corners)
    o_am.set_curr_img(frame2)
    trk.update()
    return trk.trace()[0]["H"]


@pytest.mark.parametrize("case", NT_CASES, ids=lambda c: "sm%d-am%d-ssm%d-%s" % (c[0], c[1], c[2], "_".join("%s%s" % kv for kv in c[4].items())))
def test_interface_level_sm_matches_oracle_trace(oracle, gpu_ctx, frame, case):
    """The SM loop written against the AM / SSM interface only (one C-ABI call per reference virtual):
    f, g, H per iteration and the final region against the oracle's nt:: classes, NCC included."""
    sm_kind, am, ssm, res, extra = case
    rng = np.random.default_rng(29)
    centre = (250.0, 262.0)
    corners = synth.square_corners(centre[0], centre[1], float(max(res, 50)))
    p_true = synth.random_small_homography(rng, 0.3)
    frame2 = synth.warp_frame(frame, p_true, centre)
    params = dict(leven_marq=0, max_iters=6, epsilon=-1.0)
    params.update(extra)
    o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(am, res, res); o_am.set_curr_img(frame)
    otrk = oracle.Tracker(sm_kind, o_am, o_ssm, **params)
    otrk.initialize(corners)
    gpu_ctx.set_image(frame)
    nt = NTSearchMethod(gpu_ctx, sm_kind, am, ssm, res, res, 1, **params)
    nt.initialize(corners[None])
    o_am.set_curr_img(frame2); gpu_ctx.set_image(frame2)
    otrk.update()
    nt.update()
    assert otrk.status() == 0
    otrace = otrk.trace()
    assert len(otrace) == len(nt.trace) == 6
    # The first iterations run on (numerically) identical inputs.  A second-order Hessian is not definite and is far
    # worse conditioned (x^4 terms for the homography), so the first step already amplifies the FD-gradient jitter
    # between the two grids: compare iteration 0 tightly, the trajectory loosely.
    so = bool(params.get("sec_ord_hess"))
    for it in range(1 if so else 2):
        rec, got = otrace[it], nt.trace[it]
        assert abs(got["f"][0] - rec["f"]) <= 1e-7 * abs(rec["f"]), it
        assert np.linalg.norm(got["H"][0] - rec["H"]) <= 1e-5 * np.linalg.norm(rec["H"]), it
        gs = max(np.linalg.norm(rec["g"]), 1e-3 * np.sqrt(abs(np.trace(rec["H"]))))
        assert np.linalg.norm(got["g"][0] - rec["g"]) <= 1e-4 * gs, it
    if so:
        # SSD's second-order self Hessian IS its first-order one (SSDBase.h:95-98); everywhere else the term is live
        ht = params.get("hess_type", {L.SM_ESM: 2, L.SM_FCLK: 1, L.SM_ICLK: 0}[sm_kind])
        self_type = ht in ((0, 1, 2) if sm_kind == L.SM_ESM else (0, 1))
        same = np.abs(otrace[0]["H"] - oracle_first_order_H(oracle, case, frame, frame2, corners)).max() == 0
        assert same == (am == L.AM_SSD and self_type)
        assert abs(nt.trace[1]["f"][0] - otrace[1]["f"]) <= 1e-3 * abs(otrace[1]["f"])
        np.testing.assert_allclose(nt.get_region()[0], otrk.get_region(), atol=5e-2)
    else:
        np.testing.assert_allclose(nt.get_region()[0], otrk.get_region(), atol=2e-4)


@pytest.mark.parametrize("am,ssm", [(L.AM_NCC, L.SSM_AFFINE), (L.AM_SSD, L.SSM_AFFINE), (L.AM_NCC, L.SSM_HOMOGRAPHY)])
def test_grid_patch_trackers_one_launch(oracle, gpu_ctx, frame, am, ssm):
    """Config 3 (reduced to 6x6 patches): every patch's ICLK loop runs inside one kernel launch and lands
    where the oracle's per-patch nt::ICLK does."""
    rng = np.random.default_rng(41)
    centre = (256.0, 256.0)
    region = synth.square_corners(centre[0], centre[1], 300)
    p_true = synth.random_small_homography(rng, 0.25)
    frame2 = synth.warp_frame(frame, p_true, centre)
    gpu_ctx.set_image(frame)
    gt = GridTracker(gpu_ctx, grid_size=6, patch_size=25, am=am, ssm=ssm, max_iters=30, epsilon=1e-4)
    gt.initialize(region)
    patches = gt.patch_corners(region)
    gpu_ctx.set_image(frame2)
    corners, centroids = gt.update_patches()
    assert corners.shape == (36, 2, 4) and centroids.shape == (36, 2)
    for t in range(0, 36, 5):
        o_ssm = oracle.SSM(ssm, 25, 25); o_am = oracle.AM(am, 25, 25); o_am.set_curr_img(frame)
        trk = oracle.Tracker(L.SM_ICLK, o_am, o_ssm, leven_marq=0, max_iters=30, epsilon=1e-4, hess_type=0)
        trk.initialize(patches[t])
        o_am.set_curr_img(frame2)
        iters = trk.update()
        np.testing.assert_allclose(corners[t], trk.get_region(), atol=5e-4)
        assert abs(int(gt.n_iters[t]) - iters) <= 1
    # the patch centroids follow the ground-truth motion of the region
    W = synth.homography_from_state(p_true)
    c0 = patches.mean(axis=2) - np.array(centre)
    q = (W @ np.vstack([c0.T, np.ones(36)]))
    gt_c = (q[:2] / q[2]).T + np.array(centre)
    assert np.abs(centroids - gt_c).max() < 0.25


@pytest.mark.parametrize("sm_kind,am", [(L.SM_ICLK, L.AM_NCC), (L.SM_ICLK, L.AM_SSD), (L.SM_FCLK, L.AM_SSD), (L.SM_ESM, L.AM_NCC)])
def test_track_region_equals_set_region_then_track(gpu_ctx, frame, sm_kind, am):
    """mtfhip_batch_track_region (one staged upload per frame) against the two calls it replaces, frame after frame:
    the same bits -- for the search methods that keep their template Jacobian (folded upload) and for ESM (which refreshes it)."""
    rng = np.random.default_rng(43)
    centre = (256.0, 256.0)
    region = synth.square_corners(centre[0], centre[1], 280)
    gpu_ctx.set_image(frame)
    a = GridTracker(gpu_ctx, grid_size=4, patch_size=25, am=am, ssm=L.SSM_AFFINE, max_iters=12, epsilon=1e-4)
    b = GridTracker(gpu_ctx, grid_size=4, patch_size=25, am=am, ssm=L.SSM_AFFINE, max_iters=12, epsilon=1e-4)
    for g in (a, b):
        g.tracker.sm.sm = sm_kind
        g.initialize(region)
    patches = a.patch_corners(region)
    for k in range(3):
        frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.2), centre)
        gpu_ctx.set_image(frame2)
        a.tracker.set_region(patches)
        ca, _ = a.update_patches()
        cb, _ = b.update_patches(region if k % 2 == 0 else patches)
        assert np.array_equal(ca, cb)
        assert np.array_equal(a.n_iters, b.n_iters)
        assert np.array_equal(a.tracker.get_region(), b.tracker.get_region())
        assert np.array_equal(a.tracker.batch.get_state(), b.tracker.batch.get_state())


def test_copy_and_sync_fallback_gives_the_same_results(gpu_ctx, frame, monkeypatch):
    """MTFHIP_ZERO_COPY=0 (read at batch creation): state slabs go up with hipMemcpyAsync and results come back by copy + stream
    synchronisation instead of kernels reading / writing pinned host memory -- the grid frame, the device-side loop and the
    particle filter give the same results either way."""
    rng = np.random.default_rng(47)
    centre = (256.0, 256.0)
    region = synth.square_corners(centre[0], centre[1], 280)
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.2), centre)
    normals, uniforms = rng.normal(size=(300, 8)), rng.uniform(size=300)
    out = {}
    for zc in ("1", "0"):
        monkeypatch.setenv("MTFHIP_ZERO_COPY", zc)
        gpu_ctx.set_image(frame)
        g = GridTracker(gpu_ctx, grid_size=4, patch_size=25, max_iters=10, epsilon=1e-4)
        g.initialize(region)
        lk = LKTracker(gpu_ctx, L.SM_ESM, L.SSM_HOMOGRAPHY, 60, 60, 3, host_solve=False, max_iters=8, epsilon=1e-6, materialize=0, leven_marq=1)
        lk.initialize(np.stack([synth.square_corners(200 + 40 * k, 230 + 20 * k, 70) for k in range(3)]))
        pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=300, ssm_sigma=(0.004, 0.004, 0.8, 0.004, 0.004, 0.8, 2e-6, 2e-6),
                            likelihood_alpha=5.0, mean_type=1)
        pf.initialize(synth.square_corners(250.0, 240.0, 60)[None])
        gpu_ctx.set_image(frame2)
        c1, _ = g.update_patches(region)
        c2, _ = g.update_patches()
        l1 = lk.update().copy()
        lk.set_region(lk.get_region()); l2 = lk.update().copy()
        pf.iteration(normals, uniforms)
        out[zc] = (c1, c2, l1, l2, np.array(lk.n_iters), pf.get_region().copy(), pf.batch.get_state().copy())
        pf.close(); lk.batch.close(); g.tracker.batch.close()
    # (the constant template Hessian is reduced by k_finish_host on one path and k_finish_rows on the other: two fixed summation
    # orders, so everything downstream agrees to rounding, not to the bit)
    for a, b in zip(out["1"], out["0"]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)


def test_nn_dataset_generation(gpu_ctx, frame):
    """nt::NN::generateDataset mirror: the zero perturbation reproduces the template, every row is the feature
    of its own inverse-perturbed warp, and the exhaustive search finds a stored sample at distance 0."""
    from mtf_amd.sm import NNDataset
    gpu_ctx.set_image(frame)
    ds = NNDataset(gpu_ctx, am=L.AM_SSD, resx=40, resy=40, n_samples=300, seed=5)
    corners = synth.square_corners(250, 260, 90)
    perts = np.random.default_rng(5).normal(0, 1, size=(300, 8)) * ds.sigma
    perts[0] = 0
    feats = ds.initialize(corners, perts)
    assert feats.shape == (300, 1600)
    np.testing.assert_allclose(feats[0], ds.batch.read(L.BUF_I0)[0], rtol=0, atol=1e-9)
    k, d = ds.nearest(feats[137])
    assert k == 137 and d == 0.0
    # NCC features are unit-norm and zero-mean
    dn = NNDataset(gpu_ctx, am=L.AM_NCC, resx=40, resy=40, n_samples=64, seed=6)
    fn = dn.initialize(corners)
    np.testing.assert_allclose(np.linalg.norm(fn, axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(fn.sum(axis=1), 0.0, atol=1e-10)


def test_second_order_self_hessian_ncc_not_implemented(gpu_ctx, frame):
    """NCC does not override the second-order cmptSelfHessian (AppearanceModel.h:188-191 throws); neither do we."""
    gpu_ctx.set_image(frame)
    nt = NTSearchMethod(gpu_ctx, L.SM_ESM, L.AM_NCC, L.SSM_AFFINE, 20, 20, 1, sec_ord_hess=1, hess_type=2)
    with pytest.raises(mtf_amd.FunctionNotImplemented):
        nt.initialize(synth.square_corners(250, 250, 60)[None])


@pytest.mark.parametrize("sm_kind,ssm,extra", [
    (L.SM_ESM, L.SSM_HOMOGRAPHY, dict(hess_type=5)), (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=4)), (L.SM_ESM, L.SSM_HOMOGRAPHY, dict(hess_type=3, jac_type=0)),
    (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=5, chained_warp=0)), (L.SM_FCLK, L.SSM_HOMOGRAPHY, dict(hess_type=2)),
    (L.SM_FCLK, L.SSM_AFFINE, dict(hess_type=2, chained_warp=0)), (L.SM_ICLK, L.SSM_AFFINE, dict(hess_type=2)),
    (L.SM_ICLK, L.SSM_HOMOGRAPHY, dict(hess_type=2)), (L.SM_ESM, L.SSM_HOMOGRAPHY, dict(hess_type=5, leven_marq=1))],
    ids=lambda v: "_".join("%s%s" % kv for kv in v.items()) if isinstance(v, dict) else str(v))
@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC])
def test_device_loop_second_order_hessians(oracle, gpu_ctx, frame, sm_kind, ssm, extra, am):
    """mtfhip_batch_track with sec_ord_hess (SSD, NCC): the second-order term of every search method's Hessian
    (NT/ESM.cc:315-377, NT/FCLK.cc:262-283, NT/ICLK.cc:204-252; SSDBase.cc:313-415, NCC.cc:391-410) is taken by one more pixel pass per
    iteration and the finish solves the (indefinite) system with pivoting -- final corners, iteration counts and the
    per-pass H / g / update of the trace against the oracle's second-order trackers"""
    rng = np.random.default_rng(29)
    res, B = 30, 3
    p_true = synth.random_small_homography(rng, 0.35)
    frame_b = synth.warp_frame(frame, p_true, (256.0, 256.0))
    corners = np.stack([synth.square_corners(180.0 + 70 * i, 210.0 + 30 * i, 66) + 0.25 * i for i in range(B)])
    params = dict(leven_marq=0, max_iters=12, epsilon=1e-5, sec_ord_hess=1)
    params.update(extra)
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, am, ssm, res, res, B)
    b.set_math_mode(mtf_amd.MATH_REPLAY)
    b.set_corners(corners)
    sm = mtf_amd.sm_desc(sm_kind, materialize=0, **params)
    b.init_template(sm)
    gpu_ctx.set_image(frame_b)
    b.track_trace(params["max_iters"] * 2)
    n_it, final = b.track(sm)
    recs = b.read_track_trace(n_it)
    rel = lambda a, r: float(np.linalg.norm(np.asarray(a) - np.asarray(r)) / max(np.linalg.norm(r), 1e-300))
    for t in range(B):
        o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(am, res, res); o_am.set_curr_img(frame)
        trk = oracle.Tracker(sm_kind, o_am, o_ssm, **params)
        trk.initialize(corners[t]); o_am.set_curr_img(frame_b)
        iters = trk.update()
        tr = trk.trace()
        np.testing.assert_allclose(final[t], trk.get_region(), rtol=0, atol=2e-4)
        assert abs(int(n_it[t]) - iters) <= 1
        # first pass: identical state -> H (with its second-order term), g and the update directly
        d0 = [r for r in recs[t] if not r["undo"]][0]
        assert d0["has_H"] and rel(d0["H"], tr[0]["H"]) < 1e-5 and rel(d0["g"], tr[0]["g"]) < 1e-5 and rel(d0["dp"], tr[0]["dp"]) < 1e-5
        # the term is not vacuous: without it the first-pass Hessian is measurably another matrix
        o_ssm1 = oracle.SSM(ssm, res, res); o_am1 = oracle.AM(am, res, res); o_am1.set_curr_img(frame)
        trk1 = oracle.Tracker(sm_kind, o_am1, o_ssm1, **dict(params, sec_ord_hess=0))
        trk1.initialize(corners[t]); o_am1.set_curr_img(frame_b); trk1.update()
        assert rel(trk1.trace()[0]["H"], tr[0]["H"]) > 1e-4   # (the parity gate above is 1e-5)
    b.track_trace(0); b.close()
    # NCC has no second-order self Hessian (AppearanceModel.h:188-191)
    gpu_ctx.set_image(frame)
    for bad_am, kw in ((L.AM_NCC, dict(hess_type=0)),):
        trk = LKTracker(gpu_ctx, L.SM_FCLK, L.SSM_HOMOGRAPHY, 30, 30, 2, host_solve=False, am=bad_am, sec_ord_hess=1, max_iters=5, **kw)
        with pytest.raises(mtf_amd.FunctionNotImplemented):
            trk.initialize(corners[:2])
            trk.update()


@pytest.mark.gpu
@pytest.mark.parametrize("sm_kind,ssm,extra", [
    (L.SM_ESM, L.SSM_HOMOGRAPHY, dict(hess_type=5)), (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=5, chained_warp=0)),
    (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=3, jac_type=0)), (L.SM_FCLK, L.SSM_HOMOGRAPHY, dict(hess_type=2)),
    (L.SM_FCLK, L.SSM_AFFINE, dict(hess_type=2, chained_warp=0)), (L.SM_ICLK, L.SSM_AFFINE, dict(hess_type=2)),
    (L.SM_ICLK, L.SSM_HOMOGRAPHY, dict(hess_type=2)),
    # the self types (MI.cc:697-735): InitialSelf = the initial self Hessian with its second-order part, CurrentSelf, SumOfSelf
    (L.SM_ESM, L.SSM_HOMOGRAPHY, dict(hess_type=0)), (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=1)), (L.SM_ESM, L.SSM_HOMOGRAPHY, dict(hess_type=2)),
    (L.SM_FCLK, L.SSM_AFFINE, dict(hess_type=0)), (L.SM_FCLK, L.SSM_HOMOGRAPHY, dict(hess_type=1)), (L.SM_ICLK, L.SSM_AFFINE, dict(hess_type=0)),
    (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=1, chained_warp=0)),
    # SumOfStd keeps the materialising passes (two Hessian passes); `replay`: the materialising passes for a type the recompute form also has
    (L.SM_ESM, L.SSM_HOMOGRAPHY, dict(hess_type=4)), (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=4, chained_warp=0)),
    (L.SM_FCLK, L.SSM_HOMOGRAPHY, dict(hess_type=2, replay=1)), (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=1, replay=1))],
    ids=lambda v: "_".join("%s%s" % kv for kv in v.items()) if isinstance(v, dict) else str(v))
def test_device_loop_second_order_hessians_mi(oracle, gpu_ctx, frame, sm_kind, ssm, extra):
    """sec_ord_hess with MI in mtfhip_batch_iterate / mtfhip_batch_track (MI.cc:659-695: cmptInitHessian / cmptCurrHessian + sum_p df_dI(p)
    d2I_dp2(p), the Std Hessian types of the three search methods): one more pixel pass per iteration weights the pixel-Hessian
    blocks with MI's own per-pixel gradients, taken from the iteration's gradient-factor tables -- first-pass H / g / update and
    the final region against the oracle's second-order trackers; the term is not vacuous."""
    rng = np.random.default_rng(31)
    res, B = 40, 2
    p_true = synth.random_small_homography(rng, 0.25)
    frame_b = synth.warp_frame(frame, p_true, (256.0, 256.0))
    corners = np.stack([synth.square_corners(200.0 + 60 * i, 230.0 + 25 * i, 90) + 0.25 * i for i in range(B)])
    params = dict(leven_marq=0, max_iters=10, epsilon=1e-5, sec_ord_hess=1)
    params.update(extra)
    replay = params.pop("replay", 0)
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, L.AM_MI, ssm, res, res, B)
    if replay:
        b.set_math_mode(mtf_amd.MATH_REPLAY)
    b.set_corners(corners)
    sm = mtf_amd.sm_desc(sm_kind, materialize=0, **params)
    b.init_template(sm)
    gpu_ctx.set_image(frame_b)
    f, g, H = b.iterate(sm)
    b.track_trace(params["max_iters"] * 2)
    n_it, final = b.track(sm)
    recs = b.read_track_trace(n_it)
    rel = lambda a, r: float(np.linalg.norm(np.asarray(a) - np.asarray(r)) / max(np.linalg.norm(r), 1e-300))
    S = b.S
    for t in range(B):
        o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(L.AM_MI, res, res); o_am.set_curr_img(frame)
        trk = oracle.Tracker(sm_kind, o_am, o_ssm, **params)
        trk.initialize(corners[t]); o_am.set_curr_img(frame_b)
        iters = trk.update()
        tr = trk.trace()
        d0 = recs[t][0]
        assert d0["has_H"] and rel(d0["H"], tr[0]["H"]) < 2e-5 and rel(d0["g"], tr[0]["g"]) < 2e-5 and rel(d0["dp"], tr[0]["dp"]) < 1e-4, \
            (rel(d0["H"], tr[0]["H"]), rel(d0["g"], tr[0]["g"]), rel(d0["dp"], tr[0]["dp"]))
        assert rel(H[t].reshape(S, S), tr[0]["H"]) < 2e-5 and rel(g[t], tr[0]["g"]) < 2e-5      # iterate: the same first pass
        o_ssm1 = oracle.SSM(ssm, res, res); o_am1 = oracle.AM(L.AM_MI, res, res); o_am1.set_curr_img(frame)
        trk1 = oracle.Tracker(sm_kind, o_am1, o_ssm1, **dict(params, sec_ord_hess=0))
        trk1.initialize(corners[t]); o_am1.set_curr_img(frame_b); trk1.update()
        assert rel(trk1.trace()[0]["H"], tr[0]["H"]) > 1e-4
        # (MI's loops are not contractions everywhere: the final region is compared where the oracle itself converged)
        if iters < params["max_iters"]:
            np.testing.assert_allclose(final[t], trk.get_region(), rtol=0, atol=5e-3)
    b.track_trace(0); b.close()


@pytest.mark.parametrize("n_bins,pou", [(10, 1), (10, 0), (6, 0)])
@pytest.mark.parametrize("sm_kind,ssm,extra", [(L.SM_ESM, L.SSM_HOMOGRAPHY, dict()), (L.SM_FCLK, L.SSM_AFFINE, dict()), (L.SM_ICLK, L.SSM_HOMOGRAPHY, dict()),
                                               (L.SM_ESM, L.SSM_AFFINE, dict(hess_type=0, leven_marq=1))])
def test_mi_shipped_bin_count_device_loop_and_candidates(oracle, gpu_ctx, frame, n_bins, pou, sm_kind, ssm, extra):
    """r06: MI as Config/modules.cfg ships it (mi_n_bins 10, mi_pou 1) and other counts up to ten through the device-side loop
    (mtfhip_batch_track: the recompute passes' NB = 10 kernels + k_mi_finish_fast<10>) -- per-pass H / g / update of the first pass and
    the final region against the oracle's trackers -- and through the candidate scorer (k_mi_pass_hist<.., CAND, NB = 10> + the
    generic k_mi_cand_score) against the oracle's per-candidate loop."""
    rng = np.random.default_rng(57)
    res, B = 40, 3
    frame_b = synth.warp_frame(frame, synth.random_small_homography(rng, 0.25), (256.0, 256.0))
    corners = np.stack([synth.square_corners(190.0 + 55 * i, 225.0 + 30 * i, 90) + 0.25 * i for i in range(B)])
    params = dict(leven_marq=0, max_iters=8, epsilon=1e-5)
    params.update(extra)
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, L.AM_MI, ssm, res, res, B, mi_n_bins=n_bins, mi_pou=pou, likelihood_alpha=0.05)
    b.set_corners(corners)
    sm = mtf_amd.sm_desc(sm_kind, materialize=0, **params)
    b.init_template(sm)
    # candidates of target 0 on the template's own frame
    S = b.S
    states = synth.pf_candidate_states(rng, 70)[:, :S] if ssm == L.SSM_HOMOGRAPHY else rng.normal(size=(70, 6)) * np.array([1.5, 1.5, .01, .01, .01, .01])
    o_ssm0 = oracle.SSM(ssm, res, res); o_am0 = oracle.AM(L.AM_MI, res, res, n_bins=n_bins, pou=pou, likelihood_alpha=0.05); o_am0.set_curr_img(frame)
    o_ssm0.set_corners(corners[0]); o_am0.initialize_pix_vals(o_ssm0.get("curr_pts")); o_am0.initialize_similarity()
    lik_o, sim_o = oracle.pf_score(o_am0, o_ssm0, states)
    lik, sim = b.score_candidates(states, want_similarity=True)
    np.testing.assert_allclose(sim, sim_o, rtol=1e-8)
    np.testing.assert_allclose(lik, lik_o, rtol=1e-6, atol=1e-300)
    gpu_ctx.set_image(frame_b)
    b.track_trace(params["max_iters"] * 2)
    n_it, final = b.track(sm)
    recs = b.read_track_trace(n_it)
    rel = lambda a, r: float(np.linalg.norm(np.asarray(a) - np.asarray(r)) / max(np.linalg.norm(r), 1e-300))
    for t in range(B):
        o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(L.AM_MI, res, res, n_bins=n_bins, pou=pou); o_am.set_curr_img(frame)
        trk = oracle.Tracker(sm_kind, o_am, o_ssm, **params)
        trk.initialize(corners[t]); o_am.set_curr_img(frame_b)
        iters = trk.update()
        tr = trk.trace()
        d0 = recs[t][0]
        assert rel(d0["g"], tr[0]["g"]) < 1e-5 and rel(d0["dp"], tr[0]["dp"]) < 1e-4, (rel(d0["g"], tr[0]["g"]), rel(d0["dp"], tr[0]["dp"]))
        if d0["has_H"]:
            assert rel(d0["H"], tr[0]["H"]) < 1e-5, rel(d0["H"], tr[0]["H"])
        if iters < params["max_iters"]:
            np.testing.assert_allclose(final[t], trk.get_region(), rtol=0, atol=5e-3)
    b.track_trace(0); b.close()


@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC, L.AM_MI])
@pytest.mark.parametrize("math", [mtf_amd.MATH_REPLAY, mtf_amd.MATH_FAST])
def test_multichannel_candidate_scores_and_particle_filter(oracle, gpu_ctx, am, math):
    """The candidate scorer and the particle filter over MCSSD / MCNCC / MCMI (n_channels = 3, 32FC3 frame): one row per (pixel,
    channel), the pixel's grid point shared by its three rows, mc::PixVal's interpolation order -- candidate likelihoods and
    similarities against the oracle's per-candidate loop, then two filter iterations against its nt::PF restatement on shared
    draws"""
    rng = np.random.default_rng(41)
    res, n = 20, 300
    frame = synth.make_frame_mc(256, 256)
    corners = synth.square_corners(128.0, 120.0, 70.0) + rng.uniform(-1, 1, size=(2, 4))
    alpha = {L.AM_SSD: 5.0, L.AM_NCC: 500.0, L.AM_MI: 0.05}[am]   # MI is O(1): exp(-alpha (1 / f - 1)^2) underflows for larger alpha
    mi = am == L.AM_MI
    o_ssm = oracle.SSM(0, res, res); o_am = oracle.AM(am, res, res, likelihood_alpha=alpha)
    o_am.set_channels(3); o_ssm.set_channels(3)
    o_am.set_curr_img(frame)
    o_ssm.set_corners(corners); o_am.initialize_pix_vals(o_ssm.get("curr_pts")); o_am.initialize_similarity()
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, res, res, n_particles=n, am=am, likelihood_alpha=alpha, n_channels=3,
                        ssm_sigma=(1.0, 0.6, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, dynamic_model=1, mean_type=1)
    pf.batch.set_math_mode(math)
    pf.initialize(corners[None])
    assert pf.batch.N == 3 * res * res
    states = synth.pf_candidate_states(rng, 130)
    lik_o, sim_o = oracle.pf_score(o_am, o_ssm, states)
    lik, sim = pf.batch.score_candidates(states, want_similarity=True)
    np.testing.assert_allclose(sim, sim_o, rtol=1e-9)
    np.testing.assert_allclose(lik, lik_o, rtol=1e-9 if not mi else 1e-7, atol=1e-300)   # (MI: exp(-alpha (1 / f - 1)^2) amplifies f's 1e-12)
    frame_b = synth.warp_frame(frame, np.array([0, 0, 1.1, 0, 0, -0.7, 0, 0]), (128.0, 120.0))
    o_am.set_curr_img(frame_b); gpu_ctx.set_image(frame_b)
    pp = oracle.pf_params(n, dynamic_model=1, update_type=1, mean_type=1, corner_based_sampling=1, sigma=(1.0, 0.6, 1, 1, 1, 1, 1, 1))
    o_ssm.set_corners(corners)
    st_o, ar_o = np.zeros((n, 8)), np.zeros((n, 8))
    for it in range(2):
        normals, uniforms = rng.normal(size=(n, 10)), rng.uniform(size=n)
        st_o, ar_o, w_o, ids_o, _ = oracle.pf_iteration(o_am, o_ssm, pp, st_o, ar_o, normals, uniforms, pf.max_similarity)
        pf.iteration(normals, uniforms)
        st_d, ar_d, w_d, ids_d = pf.particles()
        np.testing.assert_allclose(w_d, w_o, rtol=1e-9 if not mi else 1e-6, atol=1e-300)
        same = ids_d == ids_o
        assert same.mean() > 0.99
        np.testing.assert_allclose(st_d[same], st_o[same], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(pf.batch.get_state()[0], o_ssm.get("state"), rtol=1e-6, atol=1e-9)
    pf.close()


# ------------------------------------------------------------------ multi-channel appearance models (mc::)
MC_CASES = [
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 30, dict()),                                  # MCSSD
    (L.SM_FCLK, L.AM_SSD, L.SSM_AFFINE, 30, dict(chained_warp=0, hess_type=2)),
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, 25, dict()),                                     # MCNCC
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 30, dict(hess_type=4)),
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 30, dict()),                                   # MCMI
    (L.SM_FCLK, L.AM_MI, L.SSM_AFFINE, 30, dict()),
    (L.SM_ICLK, L.AM_MI, L.SSM_HOMOGRAPHY, 30, dict()),
    (L.SM_ESM, L.AM_MI, L.SSM_AFFINE, 30, dict(hess_type=5)),                            # Std: the dense (It, I0) Hessian form
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 30, dict(hess_type=3, jac_type=0)),            # Original: mean Jacobian rows
    (L.SM_ICLK, L.AM_MI, L.SSM_AFFINE, 30, dict(hess_type=2, chained_warp=0)),           # Std on the template's rows
    (L.SM_ESM, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=5)),           # second order over (pixel, channel) rows
    (L.SM_ICLK, L.AM_SSD, L.SSM_HOMOGRAPHY, 30, dict(sec_ord_hess=1, hess_type=2, chained_warp=0)),
]


@pytest.mark.parametrize("case", MC_CASES, ids=lambda c: "sm%d-am%d-ssm%d-%s" % (c[0], c[1], c[2], "_".join("%s%s" % kv for kv in c[4].items())))
def test_multichannel_models_match_oracle(oracle, gpu_ctx, case):
    """MCSSD / MCNCC / MCMI (the single-channel classes built with n_channels = 3, AM/src/MCSSD.cc etc.) on a 32FC3 frame:
    per-channel bilinear sampling with the mc:: operation order (imgUtils.h:505-551), one Jacobian row / Hessian block per
    (pixel, channel), against the oracle's trackers."""
    sm_kind, am, ssm, res, extra = case
    rng = np.random.default_rng(37)
    centre = (128.0, 120.0)
    frame = synth.make_frame_mc(256, 256)
    corners = synth.square_corners(centre[0], centre[1], 70.0)
    p_true = synth.random_small_homography(rng, 0.3)
    frame2 = synth.warp_frame(frame, p_true, centre)
    params = dict(leven_marq=0, max_iters=5, epsilon=-1.0)
    params.update(extra)
    o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(am, res, res)
    o_am.set_channels(3); o_ssm.set_channels(3)
    o_am.set_curr_img(frame)
    otrk = oracle.Tracker(sm_kind, o_am, o_ssm, **params)
    otrk.initialize(corners)
    gpu_ctx.set_image(frame)
    nt = NTSearchMethod(gpu_ctx, sm_kind, am, ssm, res, res, 1, am_params=dict(n_channels=3), **params)
    assert nt.batch.N == 3 * res * res and nt.batch.NP == res * res
    nt.initialize(corners[None])
    # template samples are bit-identical (same grid to 1e-13, hess/grad aside): interleaved per pixel, channel fastest
    np.testing.assert_allclose(nt.batch.read(L.BUF_I0)[0], o_am.get("I0"), rtol=0, atol=1e-9)
    o_am.set_curr_img(frame2); gpu_ctx.set_image(frame2)
    otrk.update(); nt.update()
    assert otrk.status() == 0
    otrace = otrk.trace()
    rec, got = otrace[0], nt.trace[0]
    assert abs(got["f"][0] - rec["f"]) <= 1e-7 * abs(rec["f"])
    assert np.linalg.norm(got["H"][0] - rec["H"]) <= 1e-5 * np.linalg.norm(rec["H"])
    gs = max(np.linalg.norm(rec["g"]), 1e-3 * np.sqrt(abs(np.trace(rec["H"]))))
    assert np.linalg.norm(got["g"][0] - rec["g"]) <= 1e-4 * gs
    so = bool(params.get("sec_ord_hess"))
    np.testing.assert_allclose(nt.get_region()[0], otrk.get_region(), atol=5e-2 if so else 5e-4)
    # a single-channel frame is refused, as ImageBase::setCurrImg does on a type mismatch
    gpu_ctx.set_image(np.ascontiguousarray(frame[..., 0]))
    with pytest.raises(mtf_amd.InvalidArgument):
        nt.batch.update_pix_vals()
    # the fused iteration (k_fused_mc: one launch per iteration over (pixel, channel) rows; MCMI: that launch + the MI kernels over
    # the same rows) gives the same numbers as the call-by-call path above, through batch.iterate + host solve and through the
    # device-side loop (batch.track); the second-order Hessians stay with the per-function entry points and say so
    fused_ok = not so
    mi = am == L.AM_MI
    for host_solve in (True, False):
        gpu_ctx.set_image(frame)
        lk = LKTracker(gpu_ctx, sm_kind, ssm, res, res, 1, host_solve=host_solve, am=am, am_params=dict(n_channels=3), **params)
        lk.initialize(corners[None])
        gpu_ctx.set_image(frame2)
        if not fused_ok:
            with pytest.raises(mtf_amd.FunctionNotImplemented):
                lk.update()
            lk.batch.close()
            continue
        if host_solve:
            f, g, H = lk.batch.iterate(lk.sm)
            assert abs(f[0] - rec["f"]) <= 1e-7 * abs(rec["f"])
            assert np.linalg.norm(H[0] - rec["H"]) <= (1e-5 if not mi else 1e-4) * np.linalg.norm(rec["H"])
            assert np.linalg.norm(g[0] - rec["g"]) <= 1e-4 * gs
            # the interface-visible arrays of a materialising launch are the bits of the per-function kernels
            nt2 = NTSearchMethod(gpu_ctx, sm_kind, am, ssm, res, res, 1, am_params=dict(n_channels=3), **params)
            gpu_ctx.set_image(frame); nt2.initialize(corners[None]); gpu_ctx.set_image(frame2)
            nt2.batch.update_pix_vals()
            assert np.array_equal(lk.batch.read(L.BUF_IT), nt2.batch.read(L.BUF_IT))
            nt2.batch.close()
        out = lk.update()
        np.testing.assert_allclose(out[0], otrk.get_region(), atol=5e-4 if not mi else 5e-3)
        lk.batch.close()
        if not host_solve:
            # nothing materialised: the tolerance-mode multi-channel kernels (k_fused_mc_fast; MCMI: the recompute passes over
            # (pixel, channel) rows) and their replay twins
            for math in (mtf_amd.MATH_FAST, mtf_amd.MATH_REPLAY):
                gpu_ctx.set_image(frame)
                lean = LKTracker(gpu_ctx, sm_kind, ssm, res, res, 1, host_solve=False, am=am, am_params=dict(n_channels=3), materialize=0, **params)
                lean.batch.set_math_mode(math)
                lean.initialize(corners[None])
                gpu_ctx.set_image(frame2)
                gpu_ctx.timing(1); gpu_ctx.timing_reset()
                lean.batch.track_trace(2 * params["max_iters"])
                np.testing.assert_allclose(lean.update()[0], otrk.get_region(), atol=5e-4 if not mi else 5e-3)
                _, n_p1 = gpu_ctx.timing_get("mi_pass1")
                gpu_ctx.timing(False)
                # first pass of the device loop (same state as the oracle's first iteration): what it solved
                d0 = lean.batch.read_track_trace(np.array([1]))[0][0]
                if d0["has_H"]:
                    assert np.linalg.norm(d0["H"] - rec["H"]) <= (2e-5 if not mi else 1e-4) * np.linalg.norm(rec["H"])
                assert np.linalg.norm(d0["g"] - rec["g"]) <= 1e-4 * gs
                assert np.linalg.norm(d0["dp"] - rec["dp"]) <= (2e-5 if not mi else 2e-4) * np.linalg.norm(rec["dp"])
                if mi and extra.get("hess_type") != 4:   # (ESM SumOfStd keeps the materialising iteration, single-channel as well)
                    assert (n_p1 > 0) == (math == mtf_amd.MATH_FAST), "MCMI tolerance mode must take the recompute passes"
                lean.batch.close()


@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC])
@pytest.mark.parametrize("sm_kind", [L.SM_ESM, L.SM_FCLK, L.SM_ICLK])
def test_interface_level_sm_several_targets_deferred_vs_eager(gpu_ctx, frame, sm_kind, am, monkeypatch):
    """B targets in one batch through the per-function entry points: the deferred-fusion path (one fused launch per
    iteration for all targets) and the call-by-call path (MTFHIP_LAZY=0) produce the same trajectories."""
    rng = np.random.default_rng(31)
    B, res = 3, 40
    centre = (250.0, 262.0)
    corners = np.stack([synth.square_corners(centre[0] + 40 * k - 40, centre[1] + 15 * k, 80.0) for k in range(B)])
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.3), centre)
    out = {}
    for lazy in ("0", "1"):
        monkeypatch.setenv("MTFHIP_LAZY", lazy)
        gpu_ctx.set_image(frame)
        nt = NTSearchMethod(gpu_ctx, sm_kind, am, L.SSM_HOMOGRAPHY, res, res, B, leven_marq=0, max_iters=5, epsilon=-1.0)
        nt.initialize(corners)
        gpu_ctx.set_image(frame2)
        gpu_ctx.timing(1); gpu_ctx.timing_reset()
        nt.update()
        _, n_fused = gpu_ctx.timing_get("fused_lk")
        gpu_ctx.timing(False)
        out[lazy] = (nt.get_region().copy(), [dict((k, v.copy()) for k, v in r.items()) for r in nt.trace], n_fused)
        nt.batch.close()
    assert out["0"][2] == 0 and out["1"][2] == 5
    # iteration 0 runs on identical inputs: sums differ in order only.  From then on a 1e-16 difference in the state moves
    # the sample points by ~1e-13 px, which the reference's 1e-8 finite difference turns into its ~5e-6 gradient noise.
    np.testing.assert_allclose(out["1"][0], out["0"][0], rtol=0, atol=1e-5)
    for it, (a, c) in enumerate(zip(out["0"][1], out["1"][1])):
        tol = 1e-10 if it == 0 else 1e-4
        for key in ("f", "g", "H") if it == 0 else ("f", "H"):   # g cancels towards 0 at convergence: noise-dominated there
            np.testing.assert_allclose(c[key], a[key], rtol=tol, atol=tol * max(1.0, np.abs(a[key]).max()), err_msg="%s %d" % (key, it))


@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC])
@pytest.mark.parametrize("lr", [0.3, -1.0], ids=["weighted", "running"])
def test_online_template_update(oracle, gpu_ctx, frame, am, lr):
    """updateModel (enable_learning of the search methods): the template after two updates, and the next fused iteration on the
    updated template, against the oracle; the C++ search methods call it through the AppearanceModel virtual."""
    rng = np.random.default_rng(67)
    res, centre = 30, (250.0, 262.0)
    corners = synth.square_corners(centre[0], centre[1], 60.0)
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.3), centre)
    params = dict(leven_marq=0, max_iters=3, epsilon=-1.0)
    o_ssm = oracle.SSM(L.SSM_HOMOGRAPHY, res, res); o_am = oracle.AM(am, res, res); o_am.set_curr_img(frame)
    otrk = oracle.Tracker(L.SM_ESM, o_am, o_ssm, **params)
    otrk.initialize(corners)
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, am, L.SSM_HOMOGRAPHY, res, res, 1)
    b.set_corners(corners[None])
    sm = mtf_amd.sm_desc(L.SM_ESM, **params)
    b.init_template(sm)
    o_am.set_curr_img(frame2); gpu_ctx.set_image(frame2)
    otrk.update()
    b.track(sm)
    np.testing.assert_allclose(b.get_corners()[0], otrk.get_region(), atol=2e-5)
    # learn from a deliberately displaced patch, twice, so that the template (and, for NCC, its mean, norm and moments) really
    # changes; both sides get the same state
    p_trk = o_ssm.get("state").copy()
    p_off = p_trk + np.array([0, 0, 2.5, 0, 0, -1.5, 0, 0])
    I0_before = o_am.get("I0").copy()
    for rnd in range(2):
        o_ssm.set_state(p_off); b.set_state(p_off[None])
        assert o_am.update_model(o_ssm.get("curr_pts"), lr)
        b.update_model(None, lr)
    assert np.abs(o_am.get("I0") - I0_before).max() > 1.0
    np.testing.assert_allclose(b.read(L.BUF_I0)[0], o_am.get("I0"), rtol=0, atol=1e-7)
    # the next iteration on the updated template, from the same state: f, g, H of both sides
    o_ssm.set_state(p_trk); b.set_state(p_trk[None])
    otrk.update()
    rec = otrk.trace()[0]
    f, g, H = b.iterate(sm)
    assert abs(f[0] - rec["f"]) <= 1e-6 * max(abs(rec["f"]), 1e-6)
    assert np.linalg.norm(H[0] - rec["H"]) <= 1e-5 * np.linalg.norm(rec["H"])
    gs = max(np.linalg.norm(rec["g"]), 1e-3 * np.sqrt(abs(np.trace(rec["H"]))))
    assert np.linalg.norm(g[0] - rec["g"]) <= 1e-4 * gs
    b.close()
    with pytest.raises(mtf_amd.FunctionNotImplemented):
        bm = mtf_amd.Batch(gpu_ctx, L.AM_MI, L.SSM_HOMOGRAPHY, res, res, 1)
        bm.set_corners(corners[None]); bm.init_template(mtf_amd.sm_desc(L.SM_ESM, **params))
        bm.update_model(None, 0.5)


@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC])
@pytest.mark.parametrize("sm_kind", [L.SM_ESM, L.SM_FCLK])
def test_single_large_target_device_loop_equals_host_loop(gpu_ctx, frame, am, sm_kind):
    """One 160 x 160 target has a hundred block rows per iteration (the batched tests have a handful): the device-side loop follows
    the same trajectory as the fused launch + host solve, iteration count included."""
    rng = np.random.default_rng(83)
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], 160)
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.3), centre)
    out = {}
    for host_solve in (True, False):
        gpu_ctx.set_image(frame)
        trk = LKTracker(gpu_ctx, sm_kind, L.SSM_HOMOGRAPHY, 160, 160, 1, host_solve=host_solve, max_iters=12, epsilon=1e-8,
                        materialize=0, am=am, leven_marq=0)
        trk.initialize(corners[None])
        gpu_ctx.set_image(frame2)
        out[host_solve] = (trk.update()[0].copy(), int(trk.n_iters[0]))
    np.testing.assert_allclose(out[False][0], out[True][0], rtol=0, atol=1e-6)
    assert abs(out[False][1] - out[True][1]) <= 1



PERSIST_CASES = [
    # sm, am, ssm, res, B, leven_marq, math
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 160, 1, 0, "replay"),
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 160, 1, 1, "fast"),
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 120, 1, 0, "fast"),
    (L.SM_FCLK, L.AM_SSD, L.SSM_AFFINE, 100, 1, 1, "replay"),
    (L.SM_FCLK, L.AM_NCC, L.SSM_HOMOGRAPHY, 160, 1, 0, "replay"),
    (L.SM_ICLK, L.AM_SSD, L.SSM_HOMOGRAPHY, 200, 1, 0, "fast"),     # (above the one-launch grid kernel's patch size)
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 50, 5, 0, "replay"),     # a few small targets: 5 x 10 workgroups resident together
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 400, 1, 0, "fast"),      # 625 rows: three rows per workgroup to fit 256 CUs
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", PERSIST_CASES, ids=lambda c: "sm%d-am%d-ssm%d-%dx%d-lm%d-%s" % c)
def test_persistent_loop_equals_two_launch_loop(gpu_ctx, case, monkeypatch):
    """mtfhip_batch_track through k_track_persist (every pass in one launch, in-kernel barrier between the pixel pass and the solve)
    against the launch-per-pass loop it replaces: same bits where the decomposition into workgroups is the same, and the
    bounded wait's give-up path (timeout forced to zero) finishes the loop with the launch-per-pass form to the same result."""
    sm_kind, am, ssm, res, B, lm, math = case
    rng = np.random.default_rng(97)
    big = synth.make_frame(768, 768)
    centre = (384.0, 384.0)
    frame2 = synth.warp_frame(big, synth.random_small_homography(rng, 0.3), centre)
    if B == 1:
        corners = synth.square_corners(centre[0], centre[1], float(res))[None]
    else:
        corners = np.stack([synth.square_corners(200.0 + 90 * k, 250.0 + 60 * k, float(res)) for k in range(B)])
    out = {}
    for mode in ("two-launch", "persist", "persist-gives-up"):
        monkeypatch.setenv("MTFHIP_PERSIST", "0" if mode == "two-launch" else "1")
        if mode == "persist-gives-up":
            monkeypatch.setenv("MTFHIP_PERSIST_TIMEOUT_US", "0")
        else:
            monkeypatch.delenv("MTFHIP_PERSIST_TIMEOUT_US", raising=False)
        gpu_ctx.set_image(big)
        trk = LKTracker(gpu_ctx, sm_kind, ssm, res, res, B, host_solve=False, max_iters=15, epsilon=1e-7, materialize=0, am=am,
                        leven_marq=lm)
        trk.batch.set_math_mode(mtf_amd.MATH_REPLAY if math == "replay" else mtf_amd.MATH_FAST)
        trk.initialize(corners)
        gpu_ctx.set_image(frame2)
        c1 = trk.update().copy()
        n1 = np.array(trk.n_iters).copy()
        c2 = trk.update().copy()          # a second call on the same batch (after a give-up it stays with the launch-per-pass loop)
        out[mode] = (c1, n1, c2, trk.batch.get_state().copy())
        trk.batch.close()
    assert out["two-launch"][1].max() > 2
    same_split = res * res <= 256 * 256 // B     # the default decomposition already fits the device
    for mode in ("persist", "persist-gives-up"):
        for a, b in zip(out[mode], out["two-launch"]):
            if same_split:
                assert np.array_equal(a, b), mode
            else:
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-7, err_msg=mode)


@pytest.mark.gpu
@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC, L.AM_MI])
@pytest.mark.parametrize("corner_based,dynamic_model,update_type,mean_type,likelihood_func,resampling_type", [
    (0, 0, 1, 0, 0, 1),    # config 4: RandomWalk + Compositional + AM likelihood + BinaryMultinomial, highest weight
    (1, 0, 1, 0, 0, 1),    # the reference's default sampler: 4-corner perturbations (parameters.h:262)
    (1, 1, 1, 1, 0, 1),    # shipped cfg: AutoRegression1 + Compositional (modules.cfg:163-166), mean of the states
    (0, 1, 0, 1, 1, 2),    # additive AR1, Gaussian likelihood, linear multinomial
    (0, 0, 0, 2, 2, 1),    # additive random walk, reciprocal likelihood, mean of the corners
    (1, 0, 1, 0, 0, 0),    # no resampling
    (1, 0, 1, 0, 0, 3),    # residual resampling (PF.cc:538-582), highest weight = the first of the sorted order
    (0, 1, 0, 1, 1, 3),    # residual + additive AR1 + Gaussian likelihood + mean of the states
])
def test_pf_iteration_matches_oracle(oracle, gpu_ctx, frame, am, corner_based, dynamic_model, update_type, mean_type, likelihood_func,
                                     resampling_type):
    """mtfhip_pf_iteration (sample generation, scoring, cumulative weights, resampling, estimate -- all on the device) against
    the oracle's restatement of one iteration of nt::PF::update's loop (NT/PF.cc:260-447) fed the SAME normal and uniform draws,
    two iterations in a row (the second starts from the resampled set and its auto-regression terms)."""
    rng = np.random.default_rng(101)
    n, res = 600, 30
    centre = (250.0, 240.0)
    corners = synth.square_corners(centre[0], centre[1], 80) + rng.uniform(-2, 2, size=(2, 4))
    sigma = (1.0, 0.6, 1, 1, 1, 1, 1, 1) if corner_based else (0.004, 0.004, 0.8, 0.004, 0.004, 0.8, 2e-6, 2e-6)
    alpha = {L.AM_SSD: 5.0, L.AM_NCC: 500.0, L.AM_MI: 0.05}[am]   # MI (8 bins, MI.cc:384-387) is O(1): exp(-alpha (1 / f - 1)^2)
    o_ssm = oracle.SSM(0, res, res); o_am = oracle.AM(am, res, res, likelihood_alpha=alpha); o_am.set_curr_img(frame)
    o_ssm.set_corners(corners); o_am.initialize_pix_vals(o_ssm.get("curr_pts")); o_am.initialize_similarity()
    pp = oracle.pf_params(n, dynamic_model=dynamic_model, update_type=update_type, likelihood_func=likelihood_func,
                          resampling_type=resampling_type, mean_type=mean_type, corner_based_sampling=corner_based, sigma=sigma,
                          measurement_sigma={L.AM_SSD: 0.4, L.AM_NCC: 0.01, L.AM_MI: 0.05}[am])
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, res, res, n_particles=n, ssm_sigma=sigma, likelihood_alpha=alpha, am=am,
                        dynamic_model=dynamic_model, update_type=update_type, likelihood_func=likelihood_func,
                        resampling_type=resampling_type, mean_type=mean_type, corner_based_sampling=corner_based,
                        measurement_sigma=pp.measurement_sigma)
    pf.initialize(corners[None])
    assert abs(pf.max_similarity - o_am.similarity) <= 1e-12 * max(1.0, abs(o_am.similarity))
    frame_b = synth.warp_frame(frame, np.array([0, 0, 1.2, 0, 0, -0.8, 0, 0]), centre)
    o_am.set_curr_img(frame_b); gpu_ctx.set_image(frame_b)
    st_o, ar_o = np.zeros((n, 8)), np.zeros((n, 8))
    nz = 10 if corner_based else 8
    for it in range(2):
        normals, uniforms = rng.normal(size=(n, nz)), rng.uniform(size=n)
        st_o, ar_o, w_o, ids_o, mx_o = oracle.pf_iteration(o_am, o_ssm, pp, st_o, ar_o, normals, uniforms, pf.max_similarity)
        pf.iteration(normals, uniforms)
        st_d, ar_d, w_d, ids_d = pf.particles()
        np.testing.assert_allclose(w_d, w_o, rtol=1e-9 if am != L.AM_MI else 1e-7, atol=1e-300)
        if resampling_type == 3:
            # deterministic given the weights: the sorted order (ties by index), round(w n) copies each, leftovers = the first
            assert np.array_equal(ids_d, ids_o)
            assert abs(w_d.sum() - 1.0) < 1e-12            # particle_wts are normalised in place (PF.cc:540)
            same = np.ones(n, dtype=bool)
        elif resampling_type:
            # the device's cumulative sum is a parallel scan: an id may differ only where a draw sits within rounding of a boundary
            cum = np.cumsum(w_o) / np.sum(w_o)
            bad = np.nonzero(ids_d != ids_o)[0]
            assert all(abs(cum[min(ids_d[k], ids_o[k])] - uniforms[k]) < 1e-12 for k in bad), bad
            same = ids_d == ids_o
        else:
            same = np.ones(n, dtype=bool)
        np.testing.assert_allclose(st_d[same], st_o[same], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ar_d[same], ar_o[same], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(pf.get_region()[0], o_ssm.get("curr_corners").reshape(4, 2).T, rtol=0, atol=1e-7)
        if mean_type == 2:   # MeanType::Corners re-bases the SSM on the mean corners (setCorners): follow it on the oracle side too
            assert np.allclose(pf.batch.get_state(), 0)
        else:
            np.testing.assert_allclose(pf.batch.get_state()[0], o_ssm.get("state"), rtol=1e-7, atol=1e-10)
    pf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 37, 256, 257, 5000, 20000, 65536, 70000, 1048576, 1100000, 4300000])
def test_pf_resampling_kernels_at_other_sizes(gpu_ctx, frame, n):
    """The chunked cumulative-weight scan (256 particles per wave, chunk totals scanned by the last workgroup to arrive) and the
    two-level multinomial search (chunk table in LDS up to 1 048 576 particles, searched in memory beyond; inside the chunk two
    rounds of independent probes up to 65 536 particles, bisection beyond) against NumPy on the device's own weights: the
    smallest index whose normalised cumulative weight reaches the draw."""
    rng = np.random.default_rng(7 + n)
    res = 12 if n < 1000000 else 4
    corners = synth.square_corners(250.0, 240.0, 60)
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, res, res, n_particles=n, ssm_sigma=(0.004, 0.004, 0.8, 0.004, 0.004, 0.8, 2e-6, 2e-6),
                        likelihood_alpha=5.0, am=L.AM_SSD, dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=1,
                        mean_type=0, corner_based_sampling=0)
    pf.initialize(corners[None])
    normals, uniforms = rng.normal(size=(n, 8)), rng.uniform(size=n)
    uniforms[:3] = [0.0, 1.0 - 1e-16, 0.5][: min(n, 3)]
    pf.iteration(normals, uniforms)
    st, ar, w, ids = pf.particles()
    assert w.shape == (n,) and np.all(w > 0)
    cum = np.cumsum(w.astype(np.longdouble))
    cum = (cum / cum[-1]).astype(np.float64)
    cum[-1] = max(cum[-1], 1.0)
    want = np.searchsorted(cum, uniforms, side="left")
    bad = np.nonzero(ids != want)[0]
    # (a parallel scan rounds differently from a running sum: an id may differ only where the draw sits within rounding of a
    # boundary -- the running sum of n terms carries up to n eps / 2 itself, so the margin grows with n; the spacing of the
    # boundaries is ~1 / n)
    tol = max(1e-12, 8 * n * 1.1e-16)
    assert all(abs(cum[min(ids[k], want[k])] - uniforms[k]) < tol for k in bad), bad[:10]
    assert len(bad) <= max(3, n // 100000)
    assert ids.min() >= 0 and ids.max() < n
    pf.close()


def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on uint64 arrays holding 32-bit words: the device generator's definition"""
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0) & MASK, np.uint64(k1) & MASK
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
    return c0, c1, c2, c3


def _philox_u2(r):
    a = (r[0] << np.uint64(21)) | (r[1] >> np.uint64(11)); b = (r[2] << np.uint64(21)) | (r[3] >> np.uint64(11))
    return (a.astype(np.float64) + 1.0) / 2.0 ** 53, (b.astype(np.float64) + 1.0) / 2.0 ** 53


def test_pf_device_generator_matches_its_definition(gpu_ctx, frame):
    """the device draws against a NumPy evaluation of their definition: Philox4x32-10 keyed by the seed with counter (particle,
    pair, iteration, tag), 53-bit uniforms in (0, 1], Box-Muller -- the normals through an additive random walk from the zero
    state with unit sigma (the states ARE the draws), the resampling uniforms through the ids they select"""
    n, seed = 30000, 0x1234567887654321
    corners = synth.square_corners(250, 240, 80)
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 12, 12, n_particles=n, seed=seed, update_type=0, dynamic_model=0, resampling_type=1,
                        ssm_sigma=(1.0,) * 8, corner_based_sampling=0)
    pf.initialize(corners[None])
    k = np.arange(n, dtype=np.uint64)
    for it in range(2):
        prev = pf.particles()[0].copy()
        pf.iteration()
        st, _, w, ids = pf.particles()
        z = np.empty((n, 8))
        for q in range(4):
            u0, u1 = _philox_u2(_philox4x32_10(k, q, it, 0x4E4F524D, seed & 0xFFFFFFFF, seed >> 32))
            rad = np.sqrt(-2.0 * np.log(u0))
            z[:, 2 * q], z[:, 2 * q + 1] = rad * np.cos(2 * np.pi * u1), rad * np.sin(2 * np.pi * u1)
        u = _philox_u2(_philox4x32_10(k, 0, it, 0x554E4946, seed & 0xFFFFFFFF, seed >> 32))[0]
        cum = np.cumsum(w.astype(np.longdouble)); cum = (cum / cum[-1]).astype(np.float64); cum[-1] = 1.0
        want = np.searchsorted(cum, u, side="left")
        bad = np.nonzero(ids != want)[0]
        assert all(abs(cum[min(ids[j], want[j])] - u[j]) < 1e-11 for j in bad) and len(bad) <= 3, bad[:10]
        prop = prev + z                       # additiveRandomWalk: state + N(0, 1) draws
        np.testing.assert_allclose(st, prop[ids], rtol=0, atol=2e-13)
    pf.close()


@pytest.mark.gpu
def test_pf_device_generator_and_comm(gpu_ctx, frame):
    """the Philox draws are a function of (seed, iteration, particle) only -- two filters with one seed produce identical particle
    sets (what lets every rank of a sharded filter regenerate them without communication) -- and are standard normal / uniform;
    the C-ABI all-gather with a one-rank communicator is a device copy"""
    import torch
    from mtf_amd.sm import Comm
    corners = synth.square_corners(250, 240, 80)
    gpu_ctx.set_image(frame)
    sets = []
    for seed in (11, 11, 12):
        pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=20000, seed=seed, update_type=0, resampling_type=0,
                            ssm_sigma=(1.0,) * 8, comm=Comm(0, 1, 0))
        pf.initialize(corners[None]); pf.iteration()
        sets.append(pf.particles()[0]); pf.close()
    assert np.array_equal(sets[0], sets[1]) and not np.array_equal(sets[0], sets[2])
    z = sets[0].ravel()      # additive random walk from the zero state with unit sigma: the states ARE the normal draws
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02 and abs((z ** 3).mean()) < 0.05 and abs((z ** 4).mean() - 3) < 0.15
    c = Comm(0, 1, 0)
    a = torch.arange(1000, dtype=torch.float64, device="cuda"); b = torch.zeros_like(a)
    c.allgather(a.data_ptr(), 1000, b.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    c.close()


def _run_ranks(world, fn):
    """fn(rank) on `world` host threads (ctypes releases the GIL inside every C-ABI call: the loopback all-gather rendezvous needs
    the ranks to be inside the library at the same time)"""
    import threading
    out, err = [None] * world, []

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:   # noqa: BLE001  (reported by the main thread)
            err.append((r, e))
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths), "a loopback rank did not come back"
    if err:
        raise err[0][1]
    return out


@pytest.mark.parametrize("world,n", [(2, 10001), (3, 10001), (8, 10001), (8, 10000), (8, 5), (3, 600)])
@pytest.mark.parametrize("cfg", [
    dict(corner_based_sampling=1, dynamic_model=0, update_type=1, mean_type=0, resampling_type=1),     # config 4
    dict(corner_based_sampling=1, dynamic_model=1, update_type=1, mean_type=1, resampling_type=2),     # shipped cfg: AR1, mean of states
    dict(corner_based_sampling=0, dynamic_model=0, update_type=0, mean_type=2, resampling_type=0, likelihood_func=2),
    # the shipped modules.cfg shape: several sampler distributions with adaptive weights + adaptive resampling (replicated on every rank)
    dict(corner_based_sampling=1, dynamic_model=1, update_type=1, mean_type=1, resampling_type=1, adaptive_resampling_thresh=0.3,
         ssm_sigma=[(1.0, 0.6), (3.0, 1.2), (0.3, 0.2)], likelihood_alpha=1.0, update_distr_wts=1),
])
@pytest.mark.parametrize("exchange", ["collective", "peer"])
def test_pf_sharded_loopback_equals_unsharded(frame, world, n, cfg, exchange):
    """The sharded filter as bench.py --workload pf --gpus N runs it -- mtfhip_pf_set_comm, block bounds with a ragged (or
    empty) last block, ONE in-place all-gather of ceil(n / world) weights per rank, replicated proposals and resampling --
    executed with world ranks as threads of this process over a loopback communicator (the exchange is a rendezvous + device
    copies; everything else is the RCCL path), three iterations with the device generator, against the unsharded filter:
    every rank must hold bit-identical weights, resample ids, particle sets and estimates.
    exchange="peer": the same with mtfhip_pf_set_exchange(PEER) -- the scoring kernel's stores into every rank's mailbox, the
    arrival counters the scan waits for, the two alternating mailbox vectors (three iterations use both)."""
    from mtf_amd.sm import Comm
    corners = synth.square_corners(250.0, 240.0, 80) + np.array([[0.3, -0.2, 0.1, 0.4], [0.2, 0.1, -0.3, 0.2]])
    sigma = (1.0, 0.6, 1, 1, 1, 1, 1, 1) if cfg["corner_based_sampling"] else (0.004, 0.004, 0.8, 0.004, 0.004, 0.8, 2e-6, 2e-6)
    frame_b = synth.warp_frame(frame, np.array([0, 0, 1.2, 0, 0, -0.8, 0, 0]), (250.0, 240.0))
    kw = dict(n_particles=n, ssm_sigma=sigma, likelihood_alpha=5.0, seed=77)
    kw.update(cfg)

    def run(comm):
        ctx = mtf_amd.Context(0)
        ctx.set_image(frame)
        pf = ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 24, 24, comm=comm, exchange=exchange if comm is not None else "collective", **kw)
        pf.initialize(corners[None])
        ctx.set_image(frame_b)
        rec = []
        for _ in range(3):
            pf.iteration()
            st, ar, w, ids = pf.particles()
            rec.append((st.copy(), ar.copy(), w.copy(), ids.copy(), pf.get_region().copy(), pf.batch.get_state().copy()))
        pf.close(); ctx.close()
        return rec
    ref = run(None)
    comms = Comm.loopback(world)
    got = _run_ranks(world, lambda r: run(comms[r]))
    for c in comms:
        c.close()
    for r in range(world):
        for it in range(3):
            for a, b, what in zip(got[r][it], ref[it], ("states", "ars", "weights", "ids", "corners", "state")):
                if what == "ids" and cfg["resampling_type"] == 0:
                    continue
                assert np.array_equal(a, b), "rank %d iteration %d: %s differ from the unsharded filter" % (r, it, what)


@pytest.mark.parametrize("am,resampling_type,n", [(L.AM_NCC, 3, 1000), (L.AM_MI, 1, 300), (L.AM_SSD, 3, 10001)])
def test_pf_peer_exchange_other_scorers_and_update(frame, am, resampling_type, n):
    """The peer-store exchange on the paths the main test does not take: the NCC scorer, the MI scorer (which does not store to the
    peers itself: a push launch of its own follows it), residual resampling (the weights are normalised in place, in the mailbox), and
    mtfhip_pf_update's back-to-back iterations (five exchanges enqueued without the host in between, both mailbox vectors reused).
    World 3 (a ragged last block) against the unsharded filter, bit for bit; and the refusals of mtfhip_pf_set_exchange."""
    from mtf_amd.sm import Comm
    corners = synth.square_corners(250.0, 240.0, 80) + np.array([[0.3, -0.2, 0.1, 0.4], [0.2, 0.1, -0.3, 0.2]])
    frame_b = synth.warp_frame(frame, np.array([0, 0, 1.2, 0, 0, -0.8, 0, 0]), (250.0, 240.0))
    kw = dict(n_particles=n, ssm_sigma=(1.0, 0.6, 1, 1, 1, 1, 1, 1), likelihood_alpha=5.0, seed=91, am=am, corner_based_sampling=1,
              resampling_type=resampling_type, max_iters=5, epsilon=-1.0)
    world = 3

    def run(comm):
        ctx = mtf_amd.Context(0)
        ctx.set_image(frame)
        pf = ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 20, 20, comm=comm, exchange="peer" if comm is not None else "collective", **kw)
        pf.initialize(corners[None])
        ctx.set_image(frame_b)
        ctx.timing(1); ctx.timing_reset()
        pf.update()
        st, ar, w, ids = pf.particles()
        if comm is not None:   # the exchange that ran was the peer one: five of them, and no all-gather
            assert ctx.timing_get("pf_peer_exchange")[1] == 5 and ctx.timing_get("pf_allgather")[1] == 0
        ctx.timing(False)
        out = (st.copy(), ar.copy(), w.copy(), ids.copy(), pf.get_region().copy())
        pf.close(); ctx.close()
        return out
    ref = run(None)
    comms = Comm.loopback(world)
    got = _run_ranks(world, lambda r: run(comms[r]))
    for c in comms:
        c.close()
    for r in range(world):
        for a, b, what in zip(got[r], ref, ("states", "ars", "weights", "ids", "corners")):
            assert np.array_equal(a, b), "rank %d: %s differ from the unsharded filter" % (r, what)
    # an unsharded filter has nobody to exchange with
    ctx = mtf_amd.Context(0)
    ctx.set_image(frame)
    with pytest.raises(mtf_amd.MtfHipError, match="sharded"):
        ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=64, seed=3, exchange="peer")
    with pytest.raises(ValueError, match="exchange"):
        ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=64, seed=3, exchange="ring")
    ctx.close()


@pytest.mark.parametrize("world,n", [(2, 2001), (4, 10000)])
def test_pf_peer_exchange_between_processes(tmp_path, world, n):
    """The cross-process half of the peer-store exchange, on one GPU: `world` PROCESSES share GPU 0 (RCCL would refuse the duplicate
    device; the communicator is a detached one), each exports its mailbox as a hipIpc handle, the handles travel through files, every
    rank maps the others' (hipIpcOpenMemHandle), and the weights then move as system-scope stores of the scoring kernel into the mapped
    mailboxes, the scans waiting on the mapped arrival counters -- three host-stepped iterations and six chained ones.  Every rank
    must end with the particle set, weights, resample ids and estimate of an unsharded filter run the same way (tests/helpers/
    pf_peer_rank.py with world 1), bit for bit.  What this leaves untested is only the link: peers on OTHER GPUs (xGMI)."""
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "pf_peer_rank.py")

    def spawn(rank, w, scratch):
        os.makedirs(scratch, exist_ok=True)
        return subprocess.Popen([sys.executable, helper, str(rank), str(w), scratch, str(n)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    ref_dir, sh_dir = str(tmp_path / "ref"), str(tmp_path / "sharded")
    procs = [spawn(0, 1, ref_dir)] + [spawn(r, world, sh_dir) for r in range(world)]
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-3000:]
    ref = np.load(os.path.join(ref_dir, "result_0.npz"))
    for r in range(world):
        got = np.load(os.path.join(sh_dir, "result_%d.npz" % r))
        for k in ref.files:
            assert np.array_equal(got[k], ref[k]), "rank %d: %s differs from the unsharded filter" % (r, k)


def test_bench_pf_peer_child_ranks(tmp_path, gpu_ctx):
    """bench.py's sharded_peer form of the pf_strong record runs every rank in a child process (`bench.py --pf-peer-child ...`, so
    that nothing the never-yet-multi-GPU exchange does can take the headline line down).  Two such children on GPU 0: both finish, time
    the same number of exchanges, and end on the estimate of the unsharded filter after the same 9 updates x 10 iterations."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    scratch = str(tmp_path)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--pf-peer-child", str(r), "2", "0", scratch, "10000", "3", "10"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    res = [json.load(open(os.path.join(scratch, "result_%d.json" % r))) for r in range(2)]
    gpu_ctx.set_image(synth.make_frame(1024, 1024))
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 50, 50, n_particles=10000, max_iters=10, seed=synth.DEFAULT_SEED, **bench.PF_FILTER_KW)
    pf.initialize(synth.square_corners(512, 512, 100)[None])
    for _ in range(9):
        pf.update()
    want = [float(v) for v in np.asarray(pf.get_region()).ravel()]
    pf.close()
    for r in range(2):
        assert res[r]["estimate"] == want and res[r]["peer_exchanges_timed"] == 30 and res[r]["allgather_ms"] == 0 and res[r]["seconds"] > 0


def test_pf_peer_exchange_gives_up_on_a_rank_that_never_arrives(tmp_path):
    """The device-side wait of the peer exchange is bounded: with one rank connected but never iterating, the other rank's scan spins
    for a couple of seconds, raises the filter's error word, and the iteration comes back as MTFHIP_ERR_HIP -- a failed rank costs
    seconds and an error, never a hung queue (what lets bench.py run the exchange's first multi-GPU measurement unattended)."""
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "pf_peer_rank.py")
    scratch = str(tmp_path / "s")
    os.makedirs(scratch)
    env = dict(os.environ, PF_PEER_TEST_IDLE_RANK="1")
    procs = [subprocess.Popen([sys.executable, helper, str(r), "2", scratch, "2000"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(2)]
    outs = []
    for p in procs:
        out, _ = p.communicate(timeout=300)
        outs.append(out)
        assert p.returncode == 0, out[-2000:]
    assert "GAVE_UP" in outs[0] and "waiting for another rank" in outs[0], outs[0][-2000:]


def test_pf_peer_exchange_refuses_mismatched_seeds_between_processes(tmp_path):
    """a detached communicator has no all-gather to compare the seeds with at mtfhip_pf_set_comm: they are compared through the
    mailboxes when the peers are connected, and a mismatch is refused on both ranks"""
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "pf_peer_rank.py")
    scratch = str(tmp_path / "s")
    os.makedirs(scratch)
    procs = []
    for r in range(2):
        env = dict(os.environ, PF_PEER_TEST_SEED=str(500 + r))
        procs.append(subprocess.Popen([sys.executable, helper, str(r), "2", scratch, "600"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode != 0 and "another seed" in out, out[-2000:]


def test_pf_peer_exchange_refuses_mismatched_particle_counts_between_processes(tmp_path):
    """r04 advisor: every rank addresses the peers' mailboxes with its OWN particle count and vector capacity -- ranks created with
    different n_particles stored outside the peer's allocation.  The mailbox header (fixed offsets) carries n and the capacity next to
    the seed; a mismatch is refused on both ranks when the peers are connected, before any weight is stored."""
    import subprocess
    import sys
    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "pf_peer_rank.py")
    scratch = str(tmp_path / "s")
    os.makedirs(scratch)
    procs = []
    for r in range(2):
        env = dict(os.environ, PF_PEER_TEST_EXTRA_PARTICLES=str(4096 * r))
        procs.append(subprocess.Popen([sys.executable, helper, str(r), "2", scratch, "600"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode != 0 and "same n_particles" in out, out[-2000:]


def test_pf_sharded_filter_refuses_mismatched_particle_counts(frame):
    """the same on the collective path: mtfhip_pf_set_comm gathers the particle counts next to the seeds"""
    from mtf_amd.sm import Comm
    world = 2
    comms = Comm.loopback(world)

    def run(r):
        ctx = mtf_amd.Context(0)
        ctx.set_image(frame)
        try:
            with pytest.raises(mtf_amd.MtfHipError, match="n_particles"):
                ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=64 + 64 * r, seed=11, comm=comms[r])
        finally:
            ctx.close()
        return True
    assert all(_run_ranks(world, run))
    for c in comms:
        c.close()


def test_pf_sharded_filter_refuses_mismatched_seeds(frame):
    """every rank of a sharded filter scores a block of ITS OWN proposals: ranks with different Philox keys would mix the weights of
    different particle sets.  mtfhip_pf_set_comm gathers the seeds once and refuses (r03 advisor finding); the Python front end refuses
    seed 0 (= draw one per process) with a communicator of more than one rank before anything is created."""
    from mtf_amd.sm import Comm
    world = 2
    comms = Comm.loopback(world)

    def run(r):
        ctx = mtf_amd.Context(0)
        ctx.set_image(frame)
        try:
            with pytest.raises(mtf_amd.MtfHipError, match="seed"):
                ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=64, seed=11 + r, comm=comms[r])
            with pytest.raises(ValueError, match="seed"):
                ParticleFilter(ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=64, seed=0, comm=comms[r])
        finally:
            ctx.close()
        return True
    assert all(_run_ranks(world, run))
    for c in comms:
        c.close()


def test_pf_shard_bounds_and_inplace_allgather(gpu_ctx):
    """mtfhip_pf_shard_bounds is the partition the sharded filter uses, and the loopback all-gather in its in-place form leaves
    the flat vector on every rank (ragged world: the last block is short, the tail of the buffer is padding)"""
    import torch
    from mtf_amd.sm import Comm
    n, world = 1003, 4
    b = [Comm.shard_bounds(n, world, r) for r in range(world)]
    m = b[0][2]
    assert m == 251 and [x[0] for x in b] == [0, 251, 502, 753] and [x[1] for x in b] == [251, 251, 251, 250]
    assert Comm.shard_bounds(5, 8, 7) == (5, 0, 1) and Comm.shard_bounds(5, 8, 2) == (2, 1, 1)
    comms = Comm.loopback(world)
    full = torch.arange(m * world, dtype=torch.float64, device="cuda") + 0.5

    def rank(r):
        buf = torch.full((m * world,), -1.0, dtype=torch.float64, device="cuda")
        lo, cnt, _ = b[r]
        buf[lo:lo + cnt] = full[lo:lo + cnt]
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        comms[r].allgather(buf.data_ptr() + 8 * r * m, m, buf.data_ptr(), st.cuda_stream)
        st.synchronize()
        return buf[:n].cpu().numpy()
    got = _run_ranks(world, rank)
    for c in comms:
        c.close()
    for r in range(world):
        assert np.array_equal(got[r], full[:n].cpu().numpy())


@pytest.mark.parametrize("cfg", [dict(dynamic_model=0, mean_type=0), dict(dynamic_model=1, mean_type=1), dict(dynamic_model=1, mean_type=2),
                                 dict(dynamic_model=0, mean_type=0, resampling_type=0, update_type=0, corner_based_sampling=0)])
def test_pf_lookahead_proposals_equal_separate_proposals(gpu_ctx, frame, cfg, monkeypatch):
    """With the device generator the selection pass of iteration t also makes the proposals of iteration t + 1
    (MTFHIP_PF_LOOKAHEAD=0: a k_pf_propose launch per iteration; MeanType::Corners always takes that form): the same particle
    sets, weights and estimates bit for bit -- also across a set_region / set_sampler / set_particles in between, which
    invalidate proposals made ahead"""
    corners = synth.square_corners(250.0, 240.0, 80)
    kw = dict(n_particles=2500, ssm_sigma=(1.0, 0.6, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, likelihood_alpha=5.0, seed=9)
    kw.update(cfg)
    if not kw["corner_based_sampling"]:
        kw["ssm_sigma"] = (0.004, 0.004, 0.8, 0.004, 0.004, 0.8, 2e-6, 2e-6)
    gpu_ctx.set_image(frame)
    rec = {}
    for la in ("1", "0"):
        monkeypatch.setenv("MTFHIP_PF_LOOKAHEAD", la)
        pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 20, 20, **kw); pf.initialize(corners[None])
        out = []
        for it in range(6):
            if it == 2:
                pf.set_region(corners + 0.75)
            if it == 3:
                L.check(L.lib().mtfhip_pf_set_sampler(pf._h, (L.C.c_double * 8)(*[1.5 * v for v in kw["ssm_sigma"]]), (L.C.c_double * 8)()))
            if it == 4:
                st, ar, _, _ = pf.particles(); pf.set_particles(st[::-1].copy(), ar[::-1].copy())
            pf.iteration()
            out.append([x.copy() for x in pf.particles()] + [pf.get_region().copy(), pf.batch.get_state().copy()])
        rec[la] = out
        pf.close()
    for it in range(6):
        for a, b in zip(rec["1"][it], rec["0"][it]):
            assert np.array_equal(a, b), it


@pytest.mark.parametrize("cfg", [dict(dynamic_model=0, mean_type=0), dict(dynamic_model=1, mean_type=1),
                                 dict(dynamic_model=0, mean_type=0, update_type=0, corner_based_sampling=0),
                                 dict(dynamic_model=0, mean_type=0, n_particles=257), dict(dynamic_model=1, mean_type=0, n_particles=16384),
                                 dict(dynamic_model=0, mean_type=0, n_particles=16385), dict(dynamic_model=0, mean_type=0, ssm=1, pt_based_sampling=1)])
def test_pf_selection_pass_that_scans_itself_and_perturbations_drawn_ahead(gpu_ctx, frame, cfg, monkeypatch):
    """Up to 16 384 particles with multinomial resampling the selection pass builds the cumulative weights in its own LDS (no k_pf_scan
    launch: MTFHIP_PF_LOCAL=0 restores it), and the perturbations of iteration t + 2 are drawn by extra workgroups of iteration t's
    selection launch (MTFHIP_PF_PERT_AHEAD=0: inside the look-ahead proposal): the same particle sets, resample ids, weights and
    estimates bit for bit in all four combinations -- also across set_region / set_sampler / set_particles, which invalidate what
    was drawn ahead, and through the chained update()"""
    corners = synth.square_corners(250.0, 240.0, 80)
    kw = dict(n_particles=2500, ssm_sigma=(1.0, 0.6, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, likelihood_alpha=5.0, seed=11)
    kw.update(cfg)
    ssm = L.SSM_AFFINE if kw.pop("ssm", 0) else L.SSM_HOMOGRAPHY
    if ssm == L.SSM_AFFINE:
        kw.pop("corner_based_sampling"); kw["ssm_sigma"] = (0.8, 0.8, 0.6, 0.6, 0.5, 0.5)
    elif not kw["corner_based_sampling"]:
        kw["ssm_sigma"] = (0.004, 0.004, 0.8, 0.004, 0.004, 0.8, 2e-6, 2e-6)
    gpu_ctx.set_image(frame)
    rec = {}
    for local, ahead in (("1", "1"), ("0", "0"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("MTFHIP_PF_LOCAL", local); monkeypatch.setenv("MTFHIP_PF_PERT_AHEAD", ahead)
        pf = ParticleFilter(gpu_ctx, ssm, 20, 20, epsilon=-1.0, max_iters=3, **kw); pf.initialize(corners[None])
        out = []
        for it in range(8):
            if it == 3:
                pf.set_region(corners + 0.75)
            if it == 4:
                L.check(L.lib().mtfhip_pf_set_sampler(pf._h, (L.C.c_double * 8)(*([1.5 * v for v in kw["ssm_sigma"]] + [0.0] * 2)[:8]), (L.C.c_double * 8)()))
            if it == 5:
                st, ar, _, _ = pf.particles(); pf.set_particles(st[::-1].copy(), ar[::-1].copy())
            if it == 6:
                pf.update()      # three iterations enqueued back to back
            else:
                pf.iteration()
            out.append([x.copy() for x in pf.particles()] + [pf.get_region().copy(), pf.batch.get_state().copy()])
        rec[(local, ahead)] = out
        pf.close()
    for key in (("0", "0"), ("1", "0"), ("0", "1")):
        for it in range(8):
            for a, b in zip(rec[("1", "1")][it], rec[key][it]):
                assert np.array_equal(a, b), (key, it)


def test_pf_update_chained_iterations_equal_single_steps(gpu_ctx, frame):
    """mtfhip_pf_update with a negative epsilon enqueues its iterations back to back and reads only the last estimate back:
    same particle set and estimate as the same number of mtfhip_pf_iteration calls"""
    corners = synth.square_corners(250.0, 240.0, 80)
    kw = dict(n_particles=3000, ssm_sigma=(1.0, 0.6, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, likelihood_alpha=5.0, seed=5,
              dynamic_model=1, mean_type=1, epsilon=-1.0, max_iters=4)
    gpu_ctx.set_image(frame)
    a = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 20, 20, **kw); a.initialize(corners[None])
    b = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 20, 20, **kw); b.initialize(corners[None])
    ca = a.update()
    assert a.n_iters == 4
    for _ in range(4):
        b.iteration()
    for x, y in zip(a.particles(), b.particles()):
        assert np.array_equal(x, y)
    assert np.array_equal(ca, b.get_region()) and np.array_equal(a.batch.get_state(), b.batch.get_state())
    a.close(); b.close()


@pytest.mark.parametrize("pt_based,dynamic_model", [(1, 0), (2, 0), (1, 1), (2, 1), (0, 1)])
def test_pf_affine_sampler_matches_oracle(oracle, gpu_ctx, frame, pt_based, dynamic_model):
    """Affine particle filter: Affine::generatePerturbation (point based 1 / 2: three canonical points disturbed + the affine
    map of the three pairs; geometric: geomToState of six draws, Affine.cc:464-503) under the compositional models, against
    the oracle's restatement fed the same draws"""
    rng = np.random.default_rng(55 + pt_based)
    n, res = 500, 24
    corners = synth.square_corners(250.0, 240.0, 70) + rng.uniform(-1.5, 1.5, size=(2, 4))
    if pt_based:
        sigma, mean = (0.8, 0.5, 0.7, 0.6, 0.9, 0.4, 0, 0), (0.0,) * 8
    else:   # geometric: (tx, ty, scale, theta, aspect, phi); scale and aspect are drawn around 1
        sigma, mean = (0.8, 0.6, 0.01, 0.01, 0.01, 0.05, 0, 0), (0, 0, 1.0, 0, 1.0, 0, 0, 0)
    nz = 8 if pt_based == 2 else 6
    o_ssm = oracle.SSM(1, res, res); o_am = oracle.AM(L.AM_SSD, res, res, likelihood_alpha=5.0); o_am.set_curr_img(frame)
    o_ssm.set_corners(corners); o_am.initialize_pix_vals(o_ssm.get("curr_pts")); o_am.initialize_similarity()
    pp = oracle.pf_params(n, dynamic_model=dynamic_model, update_type=1, mean_type=1, sigma=sigma, mean=mean, pt_based_sampling=pt_based)
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_AFFINE, res, res, n_particles=n, ssm_sigma=sigma, ssm_mean=mean, likelihood_alpha=5.0,
                        dynamic_model=dynamic_model, update_type=1, mean_type=1, pt_based_sampling=pt_based)
    pf.initialize(corners[None])
    st_o, ar_o = np.zeros((n, 6)), np.zeros((n, 6))
    for it in range(2):
        normals, uniforms = rng.normal(size=(n, nz)), rng.uniform(size=n)
        st_o, ar_o, w_o, ids_o, _ = oracle.pf_iteration(o_am, o_ssm, pp, st_o, ar_o, normals, uniforms, pf.max_similarity)
        pf.iteration(normals, uniforms)
        st_d, ar_d, w_d, ids_d = pf.particles()
        np.testing.assert_allclose(w_d, w_o, rtol=1e-9, atol=1e-300)
        same = ids_d == ids_o
        cum = np.cumsum(w_o) / np.sum(w_o)
        assert all(abs(cum[min(ids_d[k], ids_o[k])] - uniforms[k]) < 1e-12 for k in np.nonzero(~same)[0])
        np.testing.assert_allclose(st_d[same], st_o[same], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(ar_d[same], ar_o[same], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(pf.batch.get_state()[0][:6], o_ssm.get("state"), rtol=1e-7, atol=1e-9)
    pf.close()


@pytest.mark.parametrize("kw,msg", [
    (dict(update_type=0, pt_based_sampling=1), "point based sampling is not implemented yet"),      # Affine.cc:509-511: the reference throws
    (dict(update_type=0, pt_based_sampling=0), "stateToGeom"),                                        # additive geometric: Eigen JacobiSVD conventions
    (dict(update_type=1, dynamic_model=0, pt_based_sampling=0), "geometric sampling is not implemented yet"),   # Affine.cc:550-552
])
def test_pf_affine_sampler_refusals(gpu_ctx, frame, kw, msg):
    gpu_ctx.set_image(frame)
    with pytest.raises(L.FunctionNotImplemented, match=msg):
        ParticleFilter(gpu_ctx, L.SSM_AFFINE, 20, 20, n_particles=100, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(n_distr=5, thresh=0.2, resampling_type=1, corner_based=1, dynamic_model=1, mean_type=1),    # the shipped Config/modules.cfg:152-176 shape
    dict(n_distr=3, thresh=0.0, resampling_type=2, corner_based=0, dynamic_model=0, mean_type=0),
    dict(n_distr=1, thresh=0.33, resampling_type=1, corner_based=1, dynamic_model=0, mean_type=0, alpha=1.0),    # adaptive resampling alone
    dict(n_distr=2, thresh=0.33, resampling_type=3, corner_based=1, dynamic_model=0, mean_type=2, alpha=1.0),    # residual resampling, obeying the verdict
    dict(n_distr=4, thresh=0.0, resampling_type=0, corner_based=1, dynamic_model=0, mean_type=1),    # no resampling: the weights still adapt
], ids=lambda c: "_".join("%s%s" % kv for kv in c.items()))
def test_pf_distribution_mixture_and_adaptive_resampling(oracle, gpu_ctx, frame, cfg):
    """Several sampler distributions whose weights follow the average particle weight each produced (PF.cc:240-269, 345-369) and
    adaptive resampling (PF.cc:114-118, 381-390) -- the settings of the shipped modules.cfg -- on the device (the scan launch takes
    the per-distribution sums and sum w^2, its last workgroup writes the next weights and the verdict, the selection pass obeys it)
    against the oracle's restatement on shared draws: distribution ids, particle weights, the next distribution weights, whether the
    iteration resampled, the particle set and the estimate, over five iterations."""
    rng = np.random.default_rng(55)
    n, res = 700, 24
    centre = (250.0, 240.0)
    corners = synth.square_corners(centre[0], centre[1], 80) + rng.uniform(-2, 2, size=(2, 4))
    cb = cfg["corner_based"]
    base = np.array([1.0, 0.6, 1, 1, 1, 1, 1, 1]) if cb else np.array([0.004, 0.004, 0.8, 0.004, 0.004, 0.8, 2e-6, 2e-6])
    scales = [1.0, 3.0, 0.3, 6.0, 0.1][:cfg["n_distr"]]
    sigmas = [list(base * s) for s in scales]
    means = [[0.0] * 8 for _ in scales]
    if len(scales) > 1:
        means[1][0] = 0.3 if cb else 0.001       # a distribution with a non-zero mean
    alpha = cfg.get("alpha", 5.0)
    o_ssm = oracle.SSM(0, res, res); o_am = oracle.AM(L.AM_SSD, res, res, likelihood_alpha=alpha); o_am.set_curr_img(frame)
    o_ssm.set_corners(corners); o_am.initialize_pix_vals(o_ssm.get("curr_pts")); o_am.initialize_similarity()
    pp = oracle.pf_params(n, dynamic_model=cfg["dynamic_model"], update_type=1, resampling_type=cfg["resampling_type"], mean_type=cfg["mean_type"],
                          corner_based_sampling=cb, sigma=sigmas[0], mean=means[0])
    mx = oracle.pf_mix(sigmas, means, update_distr_wts=1, min_distr_wt=0.1, adaptive_resampling_thresh=cfg["thresh"])
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, res, res, n_particles=n, ssm_sigma=sigmas if len(sigmas) > 1 else sigmas[0],
                        ssm_mean=means if len(means) > 1 else means[0], likelihood_alpha=alpha, dynamic_model=cfg["dynamic_model"], update_type=1,
                        resampling_type=cfg["resampling_type"], mean_type=cfg["mean_type"], corner_based_sampling=cb,
                        adaptive_resampling_thresh=cfg["thresh"], update_distr_wts=1, min_distr_wt=0.1)
    pf.initialize(corners[None])
    frame_b = synth.warp_frame(frame, np.array([0, 0, 1.2, 0, 0, -0.8, 0, 0]), centre)
    o_am.set_curr_img(frame_b); gpu_ctx.set_image(frame_b)
    st_o, ar_o = np.zeros((n, 8)), np.zeros((n, 8))
    nz = 10 if cb else 8
    seen = set()
    for it in range(5):
        normals, uniforms, du = rng.normal(size=(n, nz)), rng.uniform(size=n), rng.uniform(size=n)
        st_o, ar_o, w_o, ids_o, mx_o, dids_o, dw_o, res_o = oracle.pf_iteration_ex(o_am, o_ssm, pp, mx, st_o, ar_o, normals, uniforms, pf.max_similarity, du)
        pf.iteration(normals, uniforms, du)
        st_d, ar_d, w_d, ids_d = pf.particles()
        dw_d, dids_d, res_d = pf.distributions()
        if cfg["n_distr"] > 1:
            bad = np.nonzero(dids_d != dids_o)[0]     # a draw within rounding of a boundary of the running sums may fall either way
            assert len(bad) <= 1, bad
            np.testing.assert_allclose(dw_d, dw_o, rtol=1e-9 if len(bad) == 0 else 1e-2)
        else:
            bad = np.array([], dtype=int)
        ok = np.ones(n, dtype=bool); ok[bad] = False
        if cfg["resampling_type"] == 3 and res_o:
            np.testing.assert_allclose(w_d[ok], (w_o / 1.0)[ok], rtol=1e-9, atol=1e-300)   # (normalised in place by both)
        else:
            np.testing.assert_allclose(w_d[ok], w_o[ok], rtol=1e-9, atol=1e-300)
        assert res_d == (res_o and cfg["resampling_type"] != 0)
        seen.add(res_d)
        if len(bad) == 0:
            if not res_d or cfg["resampling_type"] == 0:
                assert np.array_equal(ids_d, np.arange(n)) or cfg["resampling_type"] == 0
                same = np.ones(n, dtype=bool)
            elif cfg["resampling_type"] == 3:
                assert np.array_equal(ids_d, ids_o); same = np.ones(n, dtype=bool)
            else:
                cum = np.cumsum(w_o) / np.sum(w_o)
                d = np.nonzero(ids_d != ids_o)[0]
                assert all(abs(cum[min(ids_d[k], ids_o[k])] - uniforms[k]) < 1e-12 for k in d), d
                same = ids_d == ids_o
            np.testing.assert_allclose(st_d[same], st_o[same], rtol=1e-8, atol=1e-11)   # (five compositional steps without resampling let the rounding of the 3 x 3 products add up)
            np.testing.assert_allclose(pf.get_region()[0], o_ssm.get("curr_corners").reshape(4, 2).T, rtol=0, atol=1e-7)
        else:
            break     # (the two filters have legitimately parted)
    if cfg["thresh"] >= 0.3 and cfg["resampling_type"] != 0 and cfg["n_distr"] == 1:
        assert seen == {True, False}, "the case is meant to see both verdicts of the adaptive test: %s" % seen
    pf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("update_type", [1, 0])
def test_pf_jacobian_as_sigma(oracle, gpu_ctx, frame, update_type):
    """jacobian_as_sigma (PF.cc:58-64, 156-165, 214-227): the sampler's sigma of every frame is the Gauss-Newton step
    -d2f_dp2^-1 df_dp -- the self Hessian of the template's pixel Jacobian and the current Jacobian, through the per-function entry
    points -- against the oracle's AM / SSM functions."""
    res = 30
    centre = (250.0, 240.0)
    corners = synth.square_corners(centre[0], centre[1], 80)
    frame_b = synth.warp_frame(frame, np.array([0, 0, 1.2, 0, 0, -0.8, 0, 0]), centre)
    o_ssm = oracle.SSM(0, res, res); o_am = oracle.AM(L.AM_SSD, res, res); o_am.set_curr_img(frame)
    o_ssm.set_corners(corners)
    pts = o_ssm.get("curr_pts")
    o_am.initialize_pix_vals(pts); o_am.initialize_similarity(); o_am.initialize_grad(); o_am.initialize_pix_grad_pts(pts)
    jac = o_ssm.cmpt_pix_jacobian if update_type == 0 else o_ssm.cmpt_warped_pix_jacobian
    H0 = o_am.cmpt_self_hessian(jac(o_am.get("dI0_dx")))
    o_am.set_curr_img(frame_b)
    o_am.update_pix_vals(pts); o_am.update_similarity(); o_am.update_curr_grad(); o_am.update_pix_grad_pts(pts)
    g = o_am.cmpt_curr_jacobian(jac(o_am.get("dIt_dx")))
    sigma_o = -np.linalg.solve(H0, g)
    gpu_ctx.set_image(frame)
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, res, res, n_particles=300, likelihood_alpha=5.0, update_type=update_type, jacobian_as_sigma=1, seed=9,
                        max_iters=2)
    pf.initialize(corners[None])
    # (the device lays out its own grid, 1e-13 px from the oracle's: the grad_eps = 1e-8 finite differences amplify that to ~1e-6 per
    # gradient -- the reference's own noise floor)
    assert np.linalg.norm(pf._d2f_dp2 - H0) <= 1e-5 * np.linalg.norm(H0)
    gpu_ctx.set_image(frame_b)
    sigma_d = pf._jacobian_sigma()[:8]
    assert np.linalg.norm(sigma_d - sigma_o) <= 1e-4 * np.linalg.norm(sigma_o)
    # one Gauss-Newton step from the template: it points along the true motion (a translation of (1.2, -0.8)), if not at it
    assert sigma_d[2] > 0.3 and sigma_d[5] < -0.2
    out = pf.update()
    assert np.all(np.isfinite(out))
    pf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ssm", [L.SSM_HOMOGRAPHY, L.SSM_AFFINE])
def test_estimate_state_sigma_and_pix_sigma_filter(oracle, gpu_ctx, frame, ssm):
    """StateSpaceModel::estimateStateSigma (ProjectiveBase.cc:201-213: pix_sigma over the mean column norms of dw/dp) against the
    oracle, at the initial region and after a state change; and nt::PF's pix_sigma route (PFParams.cc:105-116, PF.cc:142-149): one
    estimated sigma row per distribution."""
    res = 24
    corners = synth.square_corners(250.0, 240.0, 80) + np.array([[0.5, -1.0, 2.0, 0.3], [1.0, 0.2, -0.4, 0.9]])
    o_ssm = oracle.SSM(ssm, res, res); o_ssm.set_corners(corners)
    gpu_ctx.set_image(frame)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, ssm, res, res, 1)
    b.set_corners(corners[None])
    np.testing.assert_allclose(b.estimate_state_sigma(0.5)[0], o_ssm.estimate_state_sigma(0.5), rtol=1e-9)
    p = (np.array([0.01, -0.02, 1.5, 0.015, 0.01, -2.0, 2e-5, -1e-5]) if ssm == L.SSM_HOMOGRAPHY else np.array([1.5, -2.0, 0.01, -0.02, 0.015, 0.01]))
    b.set_state(p[None]); o_ssm.set_state(p)
    np.testing.assert_allclose(b.estimate_state_sigma(2.0)[0], o_ssm.estimate_state_sigma(2.0), rtol=1e-9)
    b.close()
    if ssm == L.SSM_HOMOGRAPHY:
        pf = ParticleFilter(gpu_ctx, ssm, res, res, n_particles=500, pix_sigma=(0.5, 2.0), likelihood_alpha=5.0, corner_based_sampling=0, seed=3,
                            update_distr_wts=1)
        pf.initialize(corners[None])
        o2 = oracle.SSM(ssm, res, res); o2.set_corners(corners)
        np.testing.assert_allclose(pf.state_sigma[0, :8], o2.estimate_state_sigma(0.5), rtol=1e-9)
        np.testing.assert_allclose(pf.state_sigma[1, :8], o2.estimate_state_sigma(2.0), rtol=1e-9)
        gpu_ctx.set_image(synth.warp_frame(frame, np.array([0, 0, 1.0, 0, 0, -0.6, 0, 0]), (250.0, 240.0)))
        pf.iteration()
        st, ar, w, ids = pf.particles()
        dw, dids, res_d = pf.distributions()
        assert set(np.unique(dids)) == {0, 1} and np.all(np.isfinite(st)) and abs(dw.sum() - 1.0) < 0.2
        pf.close()


def test_host_publish_protocols_under_stress_fenced_and_not(tmp_path):
    """r04 advisor: the host publishes (k_finish_host, k_publish_host, the grid kernel's publish_target, the particle filter's estimate)
    hand their results over with acknowledged write-through stores instead of release fences; MTFHIP_PUBLISH_FENCE=1 restores the
    fenced, memory-model-conforming form at run time.  tools/publish_stress.py -- hundreds of interface-mode iterations, device-loop
    calls and filter iterations, each compared bit for bit with the first of its kind -- in both forms, and both forms end on the
    same bits."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dumps = []
    for fenced in ("0", "1"):
        dump = str(tmp_path / ("d%s.npz" % fenced))
        env = dict(os.environ, MTFHIP_PUBLISH_FENCE=fenced, PUBLISH_STRESS_N="400", PUBLISH_STRESS_DUMP=dump)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "publish_stress.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=env, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:]
        assert ("fenced" if fenced == "1" else "acknowledged stores") in r.stdout
        dumps.append(np.load(dump))
    assert np.array_equal(dumps[0]["ref"], dumps[1]["ref"]) and not dumps[0]["bad"].any() and not dumps[1]["bad"].any()


@pytest.mark.parametrize("B,res", [(1, 200), (1, 50), (3, 64), (8, 40)])
@pytest.mark.parametrize("sm_kind,am,ssm,extra", [
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, dict()), (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, dict(leven_marq=1)),
    (L.SM_FCLK, L.AM_SSD, L.SSM_AFFINE, dict(chained_warp=0)), (L.SM_ICLK, L.AM_SSD, L.SSM_HOMOGRAPHY, dict()),
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, dict()), (L.SM_FCLK, L.AM_NCC, L.SSM_AFFINE, dict(hess_type=2)),
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, dict(leven_marq=1))],
    ids=lambda v: "_".join("%s%s" % kv for kv in v.items()) if isinstance(v, dict) else str(v))
@pytest.mark.parametrize("materialize", [0, 1])
def test_one_launch_per_pass_equals_two_launch_loop(gpu_ctx, frame, frame2, B, res, sm_kind, am, ssm, extra, materialize, monkeypatch):
    """r05: for small batches a pass of the device-side loop is ONE launch -- the pixel pass's last-arriving workgroup sums the partial
    rows, solves and updates (kernels_step.hip) -- against the two launches it replaces (MTFHIP_STEP=0): the same decomposition, rows,
    summation order and finish bodies, so iteration counts, corners, states and the per-pass trace records are the same BITS; and a
    second call (setRegion + update) continues from a clean arrival counter."""
    if res > 64 and ssm == L.SSM_AFFINE and am == L.AM_NCC:
        pytest.skip("covered at the smaller sizes")
    corners = np.stack([synth.square_corners(200 + 31 * t, 230 + 17 * t, float(res if res <= 64 else 200) * (1.0 if res > 64 else 1.5)) for t in range(B)])
    out = {}
    for step in ("0", "1"):
        monkeypatch.setenv("MTFHIP_STEP", step)
        gpu_ctx.set_image(frame)
        b = mtf_amd.Batch(gpu_ctx, am, ssm, res, res, B)
        b.set_corners(corners)
        params = dict(leven_marq=0, max_iters=9, epsilon=1e-5)
        params.update(extra)
        sm = mtf_amd.sm_desc(sm_kind, materialize=materialize, **params)
        b.init_template(sm)
        gpu_ctx.set_image(frame2)
        b.track_trace(12)
        n1, c1 = b.track(sm)
        tr = b.read_track_trace(n1)
        st1 = b.get_state().copy()
        b.set_region(c1 + 0.4, sm)
        n2, c2 = b.track(sm)
        out[step] = (n1.copy(), c1.copy(), st1, n2.copy(), c2.copy(), b.get_state().copy(), tr)
        if materialize and sm_kind != L.SM_ICLK:
            out[step] += (b.read(L.BUF_IT).copy(), b.read(L.BUF_JT).copy())
        b.close()
    for k in range(6):
        assert np.array_equal(out["0"][k], out["1"][k]), k
    for t in range(B):
        assert len(out["0"][6][t]) == len(out["1"][6][t])
        for ra, rb in zip(out["0"][6][t], out["1"][6][t]):
            for key in ("H", "g", "dp", "corners"):
                assert np.array_equal(ra[key], rb[key]), (t, key)
            assert ra["f"] == rb["f"] and ra["undo"] == rb["undo"]
    for k in range(7, len(out["0"])):
        assert np.array_equal(out["0"][k], out["1"][k]), k


def test_frames_beyond_the_samplers_offset_arithmetic_are_refused(gpu_ctx, frame):
    """the candidate scorer multiplies row x pitch with the 24-bit multiplier (r05) and every sampler keeps texel offsets in 32 bits: a
    borrowed frame whose pitch or height reaches 2^24, or whose extent reaches 4 GiB, is refused when it is handed over, not sampled"""
    import torch
    t = torch.zeros(64, 64, dtype=torch.float32, device="cuda:0")
    for h, w, stride in ((4, 8, 1 << 24), (1 << 24, 8, 8), (40000, 30000, 30000)):
        with pytest.raises(mtf_amd.MtfHipError, match="exceed"):
            gpu_ctx.set_image_device(t.data_ptr(), h, w, stride, keep=t)
    gpu_ctx.set_image_device(t.data_ptr(), 64, 64, keep=t)   # (a legal one is still taken)
    gpu_ctx.set_image(frame)
