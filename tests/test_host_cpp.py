"""The C++ host layer (mtf_amd/host): CPU tests of its host-only logic, GPU tests of the adapter classes
driven by the C++ nt::ESM / FCLK / ICLK loops against the oracle's trackers."""
import numpy as np
import pytest

from mtf_amd import _lib as L
import mtf_amd
from mtf_amd import host, synth


def test_host_library_builds_and_qr_matches_numpy():
    host.build()
    rng = np.random.default_rng(2)
    for n in (6, 8):
        A = rng.normal(size=(n, n)); A = -(A @ A.T) - 0.1 * np.eye(n)
        A *= np.outer(10.0 ** rng.uniform(-2, 2, n), np.ones(n)); A = 0.5 * (A + A.T)
        b = rng.normal(size=n)
        np.testing.assert_allclose(host.qr_solve(A, b), np.linalg.solve(A, b), rtol=1e-8)


def test_product_library_holds_adapters_only():
    """libmtfhost.so = the adapters (HipAM, HipSSM) + the device drivers (hip::LK, hip::PF); the restated reference callers
    (nt::ESM / FCLK / ICLK / PF) and the C wrapper live in libmtfharness.so.  A translation unit that only uses the adapters links
    against the product alone."""
    import os
    import subprocess
    import tempfile
    host.build()
    syms = subprocess.run(["nm", "-DC", host.PRODUCT_LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    for restated in ("mtf::nt::ESM::", "mtf::nt::FCLK::", "mtf::nt::ICLK::", "mtf::nt::PF::", "mtfhost_create"):
        assert restated not in syms, restated
    for product in ("mtf::hip::HipAM::updatePixVals", "mtf::hip::HipSSM::compositionalUpdate", "mtf::hip::LK::update", "mtf::hip::PF::update",
                    "mtf::utils::colPivHouseholderQrSolve"):
        assert product in syms, product
    hsyms = subprocess.run(["nm", "-DC", host.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "mtf::nt::ESM::update" in hsyms and "mtfhost_create" in hsyms
    src = """
#include "HipModels.h"
#include "DeviceLK.h"
#include "DevicePF.h"
// what SearchMethod<AM, SSM> needs of its models (SM/include/mtf/SM/SearchMethod.h:13-19): ParamType and the pointer constructors
template <class AM, class SSM> struct Holder {
  typedef typename AM::ParamType AMParams; typedef typename SSM::ParamType SSMParams;
  Holder(const AMParams *a, const SSMParams *s) : am(a), ssm(s) {}
  AM am; SSM ssm;
};
int main(int argc, char **) {
  if (argc > 99) {   // never executed here (no device): instantiation + link is the test
    auto link = std::make_shared<mtf::hip::HipLink>();
    mtf::hip::HipAM::ParamType ap; ap.link = link; mtf::hip::HipSSM::ParamType sp; sp.link = link;
    Holder<mtf::hip::HipAM, mtf::hip::HipSSM> h(&ap, &sp);
    return (int)h.am.getNPix() + (int)h.ssm.getStateSize();
  }
  return 0;
}
"""
    hdir = os.path.join(os.path.dirname(host.PRODUCT_LIB_PATH), "host")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["g++", "-std=c++17", "-I", hdir, os.path.join(d, "t.cpp"), "-o", exe, "-L", os.path.dirname(host.PRODUCT_LIB_PATH),
                        "-lmtfhost", "-lmtfhip", "-Wl,-rpath," + os.path.dirname(host.PRODUCT_LIB_PATH)], check=True)
        assert "libmtfharness" not in subprocess.run(["ldd", exe], stdout=subprocess.PIPE, text=True).stdout


def test_host_types_switch_to_eigen_when_present():
    """with <Eigen/Dense> on the include path the host layer compiles against the reference's own typedefs"""
    import os
    t = open(os.path.join(os.path.dirname(host.PRODUCT_LIB_PATH), "host", "mtf_types.h")).read()
    assert "__has_include(<Eigen/Dense>)" in t and "typedef Eigen::Matrix<double, 2, 4> CornersT" in t
    # and with OpenCV the adapters take the reference's `const cv::Mat &`
    assert "__has_include(<opencv2/core/core.hpp>)" in t and "imageView(const cv::Mat &img)" in t


def test_host_qr_matches_oracle_qr(oracle):
    rng = np.random.default_rng(3)
    A = rng.normal(size=(8, 8)); A = -(A @ A.T) - np.eye(8)
    b = rng.normal(size=8)
    np.testing.assert_allclose(host.qr_solve(A, b), oracle.colpiv_qr_solve(A, b), rtol=1e-12, atol=1e-14)


def test_host_layer_fails_loudly_without_device():
    import ctypes
    if ctypes.CDLL(L.LIB_PATH).mtfhip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(host.HostError, match="no HIP device"):
        host.CppTracker(L.SM_ESM)


def test_host_layer_argument_validation():
    import ctypes
    if ctypes.CDLL(L.LIB_PATH).mtfhip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(host.HostError, match="Invalid sampling resolution"):   # ImageBase.cc:33-35, checked before the device
        host.CppTracker(L.SM_ESM, resx=0)


CASES = [
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 45, dict()),                 # reference class defaults (LM on)
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 45, dict(leven_marq=0)),
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 40, dict(chained_warp=0, hess_type=5)),
    (L.SM_FCLK, L.AM_SSD, L.SSM_HOMOGRAPHY, 45, dict()),
    (L.SM_FCLK, L.AM_SSD, L.SSM_AFFINE, 40, dict(hess_type=0)),
    (L.SM_ICLK, L.AM_SSD, L.SSM_HOMOGRAPHY, 45, dict()),
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, 25, dict()),
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 40, dict(hess_type=4)),
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict(leven_marq=0)),
    (L.SM_FCLK, L.AM_MI, L.SSM_AFFINE, 30, dict()),
    # ESM variants Original: the SM-side mean Jacobian stays on the device through AppearanceModel::cmptMeanOf
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 40, dict(jac_type=0, hess_type=3, leven_marq=0)),
    (L.SM_ESM, L.AM_NCC, L.SSM_AFFINE, 30, dict(jac_type=0, hess_type=3)),
    # second-order Hessians
    (L.SM_ESM, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=5, leven_marq=0)),
    (L.SM_ESM, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=3, jac_type=0, leven_marq=0, chained_warp=0)),
    (L.SM_ESM, L.AM_NCC, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=4, leven_marq=0)),
    (L.SM_FCLK, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=2, leven_marq=0)),
    (L.SM_FCLK, L.AM_MI, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=0)),
    (L.SM_ICLK, L.AM_SSD, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=2, leven_marq=0)),
    (L.SM_ICLK, L.AM_MI, L.SSM_AFFINE, 30, dict(sec_ord_hess=1, hess_type=1, leven_marq=0)),
    (L.SM_ESM, L.AM_MI, L.SSM_HOMOGRAPHY, 40, dict(sec_ord_hess=1, leven_marq=0)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "sm%d-am%d-ssm%d-%s" % (c[0], c[1], c[2], "_".join("%s%s" % kv for kv in c[4].items())))
def test_cpp_trackers_match_oracle(oracle, frame, case):
    sm, am, ssm, res, extra = case
    rng = np.random.default_rng(13)
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], float(max(2 * res, 60)))
    p_true = synth.random_small_homography(rng, 0.35)
    frame2 = synth.warp_frame(frame, p_true, centre)
    params = dict(max_iters=30, epsilon=1e-6, leven_marq=1)
    params.update(extra)
    o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(am, res, res); o_am.set_curr_img(frame)
    otrk = oracle.Tracker(sm, o_am, o_ssm, **params)
    otrk.initialize(corners)
    o_am.set_curr_img(frame2)
    o_iters = otrk.update()

    trk = host.CppTracker(sm, am, ssm, res, res, **params)
    img = frame.copy()
    trk.set_image(img)
    trk.initialize(corners)
    img[:] = frame2          # the caller overwrites its buffer in place (TrackerBase.h:22-26); update() re-uploads
    out = trk.update()
    np.testing.assert_allclose(out, otrk.get_region(), atol=1e-3)
    assert abs(trk.iters - o_iters) <= 2
    W = synth.homography_from_state(p_true)
    q = W @ np.vstack([corners - np.array(centre)[:, None], np.ones(4)])
    gt = q[:2] / q[2] + np.array(centre)[:, None]
    assert np.abs(out - gt).max() < (0.1 if am != L.AM_MI else 0.6)


DEVICE_LOOP_CASES = [c for c in CASES if not c[4].get("sec_ord_hess") and not (c[0] == L.SM_ICLK and c[4].get("hess_type") == 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", DEVICE_LOOP_CASES, ids=lambda c: "sm%d-am%d-ssm%d-%s" % (c[0], c[1], c[2], "_".join("%s%s" % kv for kv in c[4].items())))
def test_cpp_device_loop_search_method(oracle, frame, case):
    """mtf::hip::LK -- the C++ search-method object whose update() is ONE call (mtfhip_batch_track: the whole loop, Levenberg-Marquardt
    included, on the device) -- against the oracle's nt:: tracker and against the literal C++ nt:: loop over the virtuals, with the
    reference's parameters (class defaults: LM on).  Two frames: the second update() starts from the first one's result."""
    sm, am, ssm, res, extra = case
    rng = np.random.default_rng(13)
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], float(max(2 * res, 60)))
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.35), centre)
    frame3 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.45), centre)
    params = dict(max_iters=30, epsilon=1e-6, leven_marq=1)
    params.update(extra)
    o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(am, res, res); o_am.set_curr_img(frame)
    otrk = oracle.Tracker(sm, o_am, o_ssm, **params)
    otrk.initialize(corners)
    dev = host.CppTracker(sm, am, ssm, res, res, device_loop=True, **params)
    lit = host.CppTracker(sm, am, ssm, res, res, **params)
    for t in (dev, lit):
        t.set_image(frame.copy()); t.initialize(corners)
    tol = 1e-3 if am != L.AM_MI else 5e-3
    for f in (frame2, frame3):
        o_am.set_curr_img(f)
        o_iters = otrk.update()
        outs = []
        for t in (dev, lit):
            t.set_image(f.copy())
            outs.append(t.update())
        np.testing.assert_allclose(outs[0], otrk.get_region(), atol=tol)
        np.testing.assert_allclose(outs[0], outs[1], atol=tol)
        assert abs(dev.iters - o_iters) <= 2
    # setRegion of the search method, then one more frame
    otrk.set_region(corners); dev.set_region(corners)
    o_am.set_curr_img(frame2); otrk.update()
    dev.set_image(frame2.copy())
    np.testing.assert_allclose(dev.update(), otrk.get_region(), atol=tol)


@pytest.mark.gpu
def test_cpp_layer_error_paths(frame):
    # NCC leaves the second-order cmptSelfHessian unimplemented (AppearanceModel.h:188-191): the exception type and
    # text cross the C ABI and come back as the reference's FunctonNotImplemented
    trk = host.CppTracker(L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 20, 20, sec_ord_hess=1)
    trk.set_image(frame)
    with pytest.raises(host.HostError, match="FunctonNotImplemented"):
        trk.initialize(synth.square_corners(200, 200, 40))
    # Affine has no cmptApproxPixHessian / cmptPixHessian; a size mismatch is an InvalidArgument


@pytest.mark.gpu
@pytest.mark.parametrize("device_loop", [False, True], ids=["virtuals", "device_loop"])
@pytest.mark.parametrize("am,sm", [(L.AM_SSD, L.SM_ESM), (L.AM_NCC, L.SM_ICLK), (L.AM_MI, L.SM_FCLK)])
def test_cpp_multichannel_trackers(oracle, am, sm, device_loop):
    """MCSSD / MCNCC / MCMI through the C++ host layer (getNChannels() = 3, getPatchSize() = 3 n_pix, 32FC3 frames): the literal
    nt:: loop over the virtuals and mtf::hip::LK (one call per update(): the fused multi-channel iteration on the device)."""
    rng = np.random.default_rng(19)
    centre = (128.0, 124.0)
    frame = synth.make_frame_mc(256, 256)
    corners = synth.square_corners(centre[0], centre[1], 70.0)
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.3), centre)
    params = dict(max_iters=25, epsilon=1e-6, leven_marq=0)
    o_ssm = oracle.SSM(L.SSM_AFFINE, 30, 30); o_am = oracle.AM(am, 30, 30)
    o_am.set_channels(3); o_ssm.set_channels(3); o_am.set_curr_img(frame)
    otrk = oracle.Tracker(sm, o_am, o_ssm, **params)
    otrk.initialize(corners)
    o_am.set_curr_img(frame2)
    o_iters = otrk.update()
    trk = host.CppTracker(sm, am, L.SSM_AFFINE, 30, 30, n_channels=3, device_loop=device_loop, **params)
    img = frame.copy()
    trk.set_image(img)
    trk.initialize(corners)
    img[:] = frame2
    out = trk.update()
    np.testing.assert_allclose(out, otrk.get_region(), atol=2e-3 if am != L.AM_MI else 5e-3)
    assert abs(trk.iters - o_iters) <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("ssm", [L.SSM_HOMOGRAPHY, L.SSM_AFFINE])
def test_cpp_ssm_algebra_virtuals(frame, ssm):
    """getIdentityWarp / composeWarps / estimateWarpFromCorners / applyWarpToCorners / additiveUpdate called through the
    StateSpaceModel base class of the C++ layer: the same numbers as the C-ABI functions (held to the oracle on the CPU by
    tests/test_abi.py)."""
    rng = np.random.default_rng(93)
    S = 8 if ssm == L.SSM_HOMOGRAPHY else 6
    scale = np.array([.02, .02, 2, .02, .02, 2, 1e-4, 1e-4]) if S == 8 else np.array([2, 2, .02, .02, .02, .02])
    p1, p2 = rng.uniform(-1, 1, S) * scale, rng.uniform(-1, 1, S) * scale
    trk = host.CppTracker(L.SM_FCLK, L.AM_SSD, ssm, 20, 20, max_iters=2)
    trk.set_image(frame)
    c0 = synth.square_corners(220, 210, 60)
    trk.initialize(c0)
    cflat = np.ascontiguousarray(c0.T).reshape(-1)
    assert np.array_equal(trk.ssm_algebra(0, n_out=S), np.zeros(S))
    np.testing.assert_array_equal(trk.ssm_algebra(1, p1, p2, n_out=S), mtf_amd.compose_warps(ssm, p1, p2))
    cout = mtf_amd.apply_warp_to_pts(ssm, c0, p2)
    np.testing.assert_array_equal(trk.ssm_algebra(3, cflat, p2, n_out=8).reshape(4, 2).T, cout)
    est = trk.ssm_algebra(2, cflat, np.ascontiguousarray(cout.T).reshape(-1), n_out=S)
    np.testing.assert_allclose(est, p2, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(trk.ssm_algebra(4, p1, n_out=S), p1, rtol=0, atol=1e-15)    # state was 0 after initialize
    np.testing.assert_allclose(trk.get_region(), mtf_amd.apply_warp_to_pts(ssm, c0, p1), rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC])
def test_cpp_search_method_learns_the_template(oracle, frame, am):
    """enable_learning: nt::ESM calls am->updateModel(ssm->getPts()) after update() (NT/ESM.cc:293-295); three frames against the
    oracle's tracker followed by the oracle's updateModel."""
    rng = np.random.default_rng(71)
    res, centre, lr = 30, (250.0, 262.0), 0.4
    corners = synth.square_corners(centre[0], centre[1], 60.0)
    o_ssm = oracle.SSM(L.SSM_HOMOGRAPHY, res, res); o_am = oracle.AM(am, res, res); o_am.set_curr_img(frame)
    otrk = oracle.Tracker(L.SM_ESM, o_am, o_ssm, leven_marq=0, max_iters=6, epsilon=1e-6)
    otrk.initialize(corners)
    trk = host.CppTracker(L.SM_ESM, am, L.SSM_HOMOGRAPHY, res, res, max_iters=6, epsilon=1e-6, leven_marq=0)
    trk.set_learning(True, lr)
    trk.set_image(frame)
    trk.initialize(corners)
    f_prev = frame
    for k in range(3):
        f_next = synth.warp_frame(f_prev, synth.random_small_homography(rng, 0.25), centre)
        o_am.set_curr_img(f_next); otrk.update(); assert o_am.update_model(o_ssm.get("curr_pts"), lr)
        trk.set_image(f_next); trk.update()
        np.testing.assert_allclose(trk.get_region(), otrk.get_region(), atol=5e-4)
        f_prev = f_next
    # NN / FLANN distance feature of the patch at the final state, through AppearanceModel::updateDistFeat
    o_am.update_pix_vals(o_ssm.get("curr_pts"))
    It = o_am.get("It")
    want = It if am == L.AM_SSD else (It - It.mean()) / np.linalg.norm(It - It.mean())
    np.testing.assert_allclose(trk.dist_feat(), want, rtol=0, atol=2e-3 if am == L.AM_SSD else 2e-5)   # states differ by the trackers' 5e-4 px



@pytest.mark.gpu
@pytest.mark.parametrize("device_filter,n", [(True, 4000), (False, 300)])
def test_cpp_particle_filter(frame, device_filter, n):
    """mtf::hip::PF (the device filter behind nt::PF's interface) and mtf::nt::PF (the literal per-particle loop over the AM /
    SSM virtuals, incl. the SSM's sampler family) follow a translation; corner based homography sampling as in the reference's
    defaults."""
    from mtf_amd import host
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], 80)
    p_true = np.array([0, 0, 3.0, 0, 0, -2.0, 0, 0])
    frame2 = synth.warp_frame(frame, p_true, centre)
    pf = host.CppParticleFilter(device_filter, resx=30, resy=30, n_particles=n, max_iters=4, epsilon=1e-9, likelihood_alpha=5.0,
                                ssm_sigma=(1.5, 0.2, 1, 1, 1, 1, 1, 1), corner_based_sampling=1, seed=5)
    pf.set_image(frame); pf.initialize(corners); pf.set_image(frame2)
    out = pf.update()
    gt = corners + np.array([[3.0], [-2.0]])
    assert np.abs(out - gt).max() < 1.2, np.abs(out - gt).max()
    assert 1 <= pf.iters <= 4


@pytest.mark.gpu
@pytest.mark.parametrize("device_filter", [True, False])
@pytest.mark.parametrize("opts", [dict(ssm_sigma=[(1.5, 0.2), (4.0, 0.6), (0.4, 0.1)], adaptive_resampling_thresh=0.2, update_distr_wts=1),     # the shipped modules.cfg shape
                                  dict(ssm_sigma=(1.5, 0.2, 1, 1, 1, 1, 1, 1), jacobian_as_sigma=1, corner_based_sampling=0),
                                  dict(pix_sigma=(0.6, 2.0), corner_based_sampling=0, update_distr_wts=1)],    # sigmas from StateSpaceModel::estimateStateSigma
                         ids=["mixture_adaptive", "jacobian_as_sigma", "pix_sigma"])
def test_cpp_particle_filter_shipped_options(frame, device_filter, opts):
    """The options of the shipped Config/modules.cfg:157-176 through the C++ search methods -- several sampler distributions with
    adaptive weights + adaptive resampling; jacobian_as_sigma (the sampler's sigma of every frame is the Gauss-Newton step, taken
    through the AM / SSM virtuals and solved on the host) -- for mtf::hip::PF (device filter) and mtf::nt::PF (literal loop): both
    follow a translation."""
    from mtf_amd import host
    centre = (256.0, 250.0)
    corners = synth.square_corners(centre[0], centre[1], 80)
    frame2 = synth.warp_frame(frame, np.array([0, 0, 3.0, 0, 0, -2.0, 0, 0]), centre)
    kw = dict(resx=30, resy=30, n_particles=400 if device_filter else 150, max_iters=4, epsilon=1e-9, likelihood_alpha=5.0, corner_based_sampling=1, seed=5)
    kw.update(opts)
    pf = host.CppParticleFilter(device_filter, **kw)
    pf.set_image(frame); pf.initialize(corners); pf.set_image(frame2)
    out = pf.update()
    gt = corners + np.array([[3.0], [-2.0]])
    if opts.get("jacobian_as_sigma"):
        # the sampler's "sigma" is then one Gauss-Newton step, projective components included: a wide cloud by construction (the value of
        # the step itself is held to the oracle in test_pf_jacobian_as_sigma) -- here: the option runs through both search methods
        assert np.all(np.isfinite(out)) and np.abs(out - gt).max() < 40
    elif "pix_sigma" in opts:
        assert np.abs(out - gt).max() < 3.0, np.abs(out - gt).max()   # an 8-dof direct-sampling cloud: coarser than the corner-based one
    else:
        assert np.abs(out - gt).max() < 1.5, np.abs(out - gt).max()


@pytest.mark.gpu
def test_several_distributions_without_update_distr_wts_refused_by_every_front_end(gpu_ctx, frame):
    """PFParams::update_distr_wts defaults to 0 in the reference, and with several sampler distributions it then zeroes the weights and
    draws every particle's distribution from an all-zero discrete distribution (NT/PF.cc:241-257).  r03 had three behaviours for
    that configuration (C ABI refused, the C++ nt::PF produced NaN probabilities, the Python wrapper silently switched the flag on):
    now one -- every front end refuses, with the reason, and nothing leaks when mtf::hip::PF's constructor throws."""
    from mtf_amd import host
    from mtf_amd.sm import ParticleFilter
    two = [(1.5, 0.2), (4.0, 0.6)]
    gpu_ctx.set_image(frame)
    with pytest.raises(mtf_amd.FunctionNotImplemented, match="update_distr_wts"):
        ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 20, 20, n_particles=64, ssm_sigma=two, corner_based_sampling=1, seed=3)
    for device_filter in (True, False):
        for _ in range(3):   # (the constructor throws after the adapters exist: repeated, it must not accumulate device filters)
            with pytest.raises(host.HostError, match="update_distr_wts"):
                host.CppParticleFilter(device_filter, resx=20, resy=20, n_particles=64, ssm_sigma=two, corner_based_sampling=1, seed=3)
    # a pix_sigma-only configuration (PF.h: "ssm_sigma is then not used") constructs without an ssm_sigma row
    pf = host.CppParticleFilter(True, resx=20, resy=20, n_particles=64, pix_sigma=(0.8,), ssm_sigma=(), corner_based_sampling=0, seed=3)
    pf.set_image(frame); pf.initialize(synth.square_corners(256.0, 250.0, 60)); pf.update()


@pytest.mark.gpu
def test_cpp_ssm_sampler_virtuals(frame):
    """StateSpaceModel::initializeSampler / compositionalRandomWalk through the base class: draws are reproducible under a seed,
    and (direct sampling, identity base state) each state component is N(0, sigma_k)"""
    from mtf_amd import host
    pf = host.CppParticleFilter(False, resx=10, resy=10, n_particles=10, corner_based_sampling=0)
    pf.set_image(frame); pf.initialize(synth.square_corners(200, 200, 60))
    sigma = (0.01, 0.02, 2.0, 0.01, 0.02, 1.0, 1e-5, 2e-5)
    a = pf.random_walk_samples(7, 20000, sigma); b = pf.random_walk_samples(7, 20000, sigma)
    assert np.array_equal(a, b)
    np.testing.assert_allclose(a.std(axis=0), sigma, rtol=0.05)
    assert np.all(np.abs(a.mean(axis=0)) < 0.05 * np.asarray(sigma))


@pytest.mark.gpu
def test_getters_are_never_stale_with_eager_getters(oracle, frame):
    """ssm->getPts() read through the base class by code that is not one of the adapters: with eager getters the bytes are the
    oracle's points after every compositionalUpdate; without, the object is only a key (documented: the mirror is refreshed by
    syncPts() or by the eager mode) -- INTEGRATION.md, "host mirrors"."""
    from mtf_amd import host
    res = 20
    corners = synth.square_corners(220, 230, 70) + np.array([[0.3, -0.2, 0.1, 0.4], [0.2, 0.1, -0.3, 0.0]])
    tr = host.CppTracker(L.SM_ESM, resx=res, resy=res, max_iters=1)
    tr.set_image(frame); tr.initialize(corners)
    o_ssm = oracle.SSM(0, res, res); o_ssm.set_corners(corners)
    rng = np.random.default_rng(3)
    for k in range(3):
        dp = synth.random_small_homography(rng, 0.3)
        o_ssm.compositional_update(dp)
        pts = tr.pts_after_update(dp, True, res * res)
        np.testing.assert_allclose(pts, o_ssm.get("curr_pts").reshape(-1, 2).T, rtol=0, atol=1e-9)
    np.testing.assert_allclose(tr.get_region(), o_ssm.get("curr_corners").reshape(4, 2).T, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("am,ssm,hess_type", [(L.AM_SSD, L.SSM_HOMOGRAPHY, 1), (L.AM_NCC, L.SSM_AFFINE, 2)])
def test_templated_search_method_shape_instantiates_the_adapters(oracle, frame, am, ssm, hess_type):
    """SearchMethod<AM, SSM> of the reference keeps its models BY VALUE and builds them from `const AM::ParamType *` /
    `const SSM::ParamType *` (SM/include/mtf/SM/SearchMethod.h:13-19): HipAM / HipSSM provide ParamType + those constructors
    (sharing one HipLink), and FCLK<HipAM, HipSSM> in that shape (harness/TemplatedSM.h) lands where the oracle's FCLK does."""
    rng = np.random.default_rng(31)
    corners = synth.square_corners(240.0, 250.0, 80)
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.4), (240.0, 250.0))
    got, n = host.templated_fclk(frame, frame2, corners, am=am, ssm=ssm, resx=40, resy=40, max_iters=15, epsilon=1e-4, hess_type=hess_type)
    o_ssm = oracle.SSM(ssm, 40, 40); o_am = oracle.AM(am, 40, 40); o_am.set_curr_img(frame)
    trk = oracle.Tracker(L.SM_FCLK, o_am, o_ssm, leven_marq=0, max_iters=15, epsilon=1e-4, hess_type=hess_type)
    trk.initialize(corners); o_am.set_curr_img(frame2)
    iters = trk.update()
    assert abs(n - iters) <= 1
    np.testing.assert_allclose(got, trk.get_region(), rtol=0, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("sm,am,ssm,jac_type,hess_type,leven_marq", [
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 1, 2, 0),      # the shipped ESM: DiffOfJacs + SumOfSelf
    (L.SM_ESM, L.AM_NCC, L.SSM_AFFINE, 0, 3, 0),          # Original + Original: the mean Jacobian
    (L.SM_ESM, L.AM_SSD, L.SSM_HOMOGRAPHY, 1, 0, 1),      # InitialSelf + Levenberg-Marquardt (init_d2f_dp2 re-read every pass)
    (L.SM_ESM, L.AM_NCC, L.SSM_HOMOGRAPHY, 1, 4, 0),      # SumOfStd
    (L.SM_ICLK, L.AM_SSD, L.SSM_HOMOGRAPHY, 1, 0, 0),     # the shipped ICLK: InitialSelf
    (L.SM_ICLK, L.AM_NCC, L.SSM_AFFINE, 1, 0, 1),         # + Levenberg-Marquardt (d2f_dp2_orig)
    (L.SM_ICLK, L.AM_SSD, L.SSM_AFFINE, 1, 1, 0),         # CurrentSelf: dIt_dpssm sized in the constructor
    (L.SM_ICLK, L.AM_NCC, L.SSM_HOMOGRAPHY, 1, 2, 0),     # Std
])
def test_templated_esm_and_iclk_shapes_over_the_adapters(oracle, frame, sm, am, ssm, jac_type, hess_type, leven_marq):
    """ESM<AM, SSM> / ICLK<AM, SSM> of the reference (SM/src/ESM.cc:14-316, ICLK.cc:13-263) -- models by value built from ParamType
    pointers, init_pix_jacobian & co. sized from am.getPatchSize() / ssm.getStateSize(), setRegion's own refresh -- instantiated over
    HipAM / HipSSM (harness/TemplatedSM.h) and run against the oracle's trackers: update(), setRegion(), update()."""
    rng = np.random.default_rng(33)
    corners = synth.square_corners(240.0, 250.0, 80)
    frame2 = synth.warp_frame(frame, synth.random_small_homography(rng, 0.4), (240.0, 250.0))
    move = (0.75, -0.5)
    got, got2, n = host.templated_sm(sm, frame, frame2, corners, am=am, ssm=ssm, resx=40, resy=40, max_iters=15, epsilon=1e-4, jac_type=jac_type,
                                     hess_type=hess_type, leven_marq=leven_marq, move=move)
    o_ssm = oracle.SSM(ssm, 40, 40); o_am = oracle.AM(am, 40, 40); o_am.set_curr_img(frame)
    trk = oracle.Tracker(sm, o_am, o_ssm, leven_marq=leven_marq, max_iters=15, epsilon=1e-4, hess_type=hess_type, jac_type=jac_type, chained_warp=1)
    trk.initialize(corners); o_am.set_curr_img(frame2)
    iters = trk.update()
    assert abs(n[0] - iters) <= 1
    np.testing.assert_allclose(got, trk.get_region(), rtol=0, atol=2e-4)
    trk.set_region(trk.get_region() + np.array(move)[:, None])
    iters2 = trk.update()
    assert abs(n[1] - iters2) <= 1
    np.testing.assert_allclose(got2, trk.get_region(), rtol=0, atol=5e-4)
