"""CPU: the C++ oracle against the committed golden fixtures (tests/golden/lk_golden.npz, produced by the
independent NumPy re-derivation oracle/numpy_ref.py via tests/golden/make_golden.py).

Tolerances: samples are bit-level (1e-12); anything downstream of the reference's grad_eps = 1e-8
finite difference carries its ~5e-6 absolute noise per gradient component, so gradients get 5e-5 abs,
Jacobian rows / g / H / dp the north-star 1e-5 relative (H) or a little looser where the quantity is
itself a cancellation (g, dp: 1e-4)."""
import os

import numpy as np
import pytest

from mtf_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden.npz"))


@pytest.fixture(scope="module")
def img():
    return synth.make_frame(*[int(v) for v in G["img_shape"]], seed=int(G["img_seed"]))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def test_sampling_and_gradient_golden(oracle, img):
    pts = G["pts"]
    flat = np.ascontiguousarray(pts.T.ravel())
    np.testing.assert_allclose(oracle.get_pix_vals(img, flat), G["pix_vals"], rtol=0, atol=1e-12)
    g = oracle.get_img_grad(img, flat).reshape(2, -1).T
    np.testing.assert_allclose(g, G["img_grad"], rtol=0, atol=5e-5)
    # constant border value
    assert G["pix_vals"][0] == 128.0 and G["pix_vals"][9] == 128.0
    for x, y, v in zip(pts[0], pts[1], G["pix_vals"]):
        assert abs(oracle.get_pix_val(img, x, y) - v) < 1e-12


@pytest.mark.parametrize("tag", ["sq", "quad"])
def test_homography_ssd_step_golden(oracle, img, tag):
    res = int(G[tag + "_res"])
    corners, p = G[tag + "_corners"], G[tag + "_p"]
    ssm = oracle.SSM(oracle.SSM_HOM, res, res)
    am = oracle.AM(oracle.AM_SSD, res, res)
    am.set_curr_img(img)
    ssm.set_corners(corners)
    np.testing.assert_allclose(ssm.get("init_pts").reshape(-1, 2).T, G[tag + "_init_pts_full"], atol=1e-9)
    pts0 = ssm.get("curr_pts")
    am.initialize_pix_vals(pts0); am.initialize_pix_grad_pts(pts0)
    am.initialize_similarity(); am.initialize_grad(); am.initialize_hess()
    np.testing.assert_allclose(am.get("I0"), G[tag + "_I0_full"], atol=1e-9)
    J0 = ssm.cmpt_warped_pix_jacobian(am.get("dI0_dx"))
    assert rel(J0.reshape(8, -1).T, G[tag + "_J0_full"]) < 1e-5
    ssm.set_state(p)
    pts = ssm.get("curr_pts")
    am.update_pix_vals(pts); am.update_similarity(False); am.update_curr_grad(); am.update_init_grad()
    am.update_pix_grad_pts(pts)
    np.testing.assert_allclose(am.get("It"), G[tag + "_It_full"], atol=1e-9)
    np.testing.assert_allclose(am.get("dIt_dx").reshape(2, -1).T, G[tag + "_grad_full"], atol=5e-5)
    Jt = ssm.cmpt_warped_pix_jacobian(am.get("dIt_dx"))
    assert rel(Jt.reshape(8, -1).T, G[tag + "_Jt_full"]) < 1e-5
    assert abs(am.similarity - float(G[tag + "_f"])) <= 1e-10 * abs(float(G[tag + "_f"]))
    # FCLK: g = df_dIt Jt, H = -Jt^T Jt
    g = am.cmpt_curr_jacobian(Jt)
    H = am.cmpt_self_hessian(Jt)
    assert rel(H, G[tag + "_fclk_H"]) < 1e-5
    assert rel(g, G[tag + "_fclk_g"]) < 1e-5
    assert rel(-oracle.colpiv_qr_solve(H, g), G[tag + "_fclk_dp"]) < 1e-5
    # ESM shipped default: DiffOfJacs + SumOfSelf
    g = 0.5 * am.cmpt_difference_of_jacobians(J0, Jt)
    H = 0.5 * (am.cmpt_self_hessian(Jt) + am.cmpt_self_hessian(J0))
    assert rel(H, G[tag + "_esm_H"]) < 1e-5
    assert rel(g, G[tag + "_esm_g"]) < 1e-5
    assert rel(-oracle.colpiv_qr_solve(H, g), G[tag + "_esm_dp"]) < 1e-5


def test_affine_ncc_golden(oracle, img):
    res = 25
    ssm = oracle.SSM(oracle.SSM_AFF, res, res)
    am = oracle.AM(oracle.AM_NCC, res, res)
    am.set_curr_img(img)
    ssm.set_corners(G["ncc_corners"])
    pts0 = ssm.get("curr_pts")
    am.initialize_pix_vals(pts0); am.initialize_pix_grad_pts(pts0)
    am.initialize_similarity(); am.initialize_grad(); am.initialize_hess()
    J0 = ssm.cmpt_warped_pix_jacobian(am.get("dI0_dx"))
    assert rel(J0.reshape(6, -1).T[:16], G["ncc_J0_head"]) < 1e-5
    ssm.set_state(G["ncc_p"])
    am.update_pix_vals(ssm.get("curr_pts"))
    am.update_similarity(False)
    am.update_init_grad(); am.update_curr_grad()
    assert abs(am.similarity - float(G["ncc_f"])) < 1e-12
    np.testing.assert_allclose(am.get("df_dIt")[:16], G["ncc_df_dIt_head"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(am.get("df_dI0")[:16], G["ncc_df_dI0_head"], rtol=1e-9, atol=1e-15)
    assert rel(am.cmpt_init_jacobian(J0), G["ncc_g_init"]) < 1e-5
    assert rel(am.cmpt_self_hessian(J0), G["ncc_H_self_J0"]) < 1e-5


def test_mi_golden(oracle, img):
    """MI (8 bins) on a 40 x 40 homography patch: similarity, df_dIt . Jt and cmptCurrHessian(Jt) of the C++ oracle against the
    NumPy re-derivation (AM/src/MI.cc:346-382, 426-442, 603-637).  The reference's truncated 2/3 (histUtils.h:11) is a 6.7e-12
    difference in the B-spline; H, g and dp are held to the north-star 1e-5."""
    res = 40
    ssm = oracle.SSM(oracle.SSM_HOM, res, res)
    am = oracle.AM(oracle.AM_MI, res, res)
    am.set_curr_img(img)
    ssm.set_corners(G["mi_corners"])
    pts0 = ssm.get("curr_pts")
    am.initialize_pix_vals(pts0); am.initialize_pix_grad_pts(pts0)
    am.initialize_similarity(); am.initialize_grad(); am.initialize_hess()
    ssm.set_state(G["mi_p"])
    pts = ssm.get("curr_pts")
    am.update_pix_vals(pts); am.update_pix_grad_pts(pts)
    am.update_similarity(False); am.update_curr_grad(); am.update_init_grad()
    np.testing.assert_allclose(am.get("It")[:16], G["mi_It_head"], atol=1e-10)
    assert abs(am.similarity - float(G["mi_f"])) < 1e-9
    np.testing.assert_allclose(am.get("df_dIt")[:16], G["mi_df_dIt_head"], rtol=1e-6, atol=1e-12)
    Jt = ssm.cmpt_warped_pix_jacobian(am.get("dIt_dx"))
    g = am.cmpt_curr_jacobian(Jt)
    H = am.cmpt_curr_hessian(Jt)
    assert rel(g, G["mi_g_curr"]) < 1e-5
    assert rel(H, G["mi_H_curr"]) < 1e-5
    assert rel(oracle.colpiv_qr_solve(H, g), np.linalg.solve(G["mi_H_curr"], G["mi_g_curr"])) < 1e-4   # (the MI Hessian amplifies: noise floor test)


def test_second_order_affine_ssd_golden(oracle, img):
    """sec_ord_hess: the nine-sample image Hessian at the warped points, the affine pixel Hessian of the chained warp and SSD's
    second-order current Hessian of the C++ oracle against the NumPy re-derivation (make_golden.py)."""
    res = 22
    ssm = oracle.SSM(oracle.SSM_AFF, res, res)
    am = oracle.AM(oracle.AM_SSD, res, res)
    am.set_curr_img(img)
    ssm.set_corners(G["so_corners"])
    pts0 = ssm.get("curr_pts")
    am.initialize_pix_vals(pts0); am.initialize_pix_grad_pts(pts0); am.initialize_pix_hess_pts(pts0)
    am.initialize_similarity(); am.initialize_grad(); am.initialize_hess()
    ssm.set_state(G["so_p"])
    pts = ssm.get("curr_pts")
    am.update_pix_vals(pts); am.update_pix_grad_pts(pts); am.update_pix_hess_pts(pts)
    am.update_similarity(False); am.update_curr_grad(); am.update_init_grad()
    np.testing.assert_allclose(am.get("d2It_dx2").reshape(-1, 4)[:16], G["so_img_hess_head"], rtol=0, atol=1e-10)
    Jt = ssm.cmpt_warped_pix_jacobian(am.get("dIt_dx"))
    Dt = ssm.cmpt_warped_pix_hessian(am.get("d2It_dx2"), am.get("dIt_dx"))
    assert rel(Dt[:8], G["so_pix_hess_head"]) < 1e-10
    assert rel(am.cmpt_curr_hessian2(Jt, Dt), G["so_H_curr2"]) < 1e-5


def test_pf_scores_golden(oracle, img):
    res = 20
    ssm = oracle.SSM(oracle.SSM_HOM, res, res)
    am = oracle.AM(oracle.AM_SSD, res, res, likelihood_alpha=1.0)
    am.set_curr_img(img)
    ssm.set_corners(G["pf_corners"])
    am.initialize_pix_vals(ssm.get("curr_pts")); am.initialize_similarity()
    lik, sim = oracle.pf_score(am, ssm, G["pf_states"])
    np.testing.assert_allclose(lik, G["pf_likelihood"], rtol=1e-9)


def test_estimate_state_sigma_golden(oracle):
    """StateSpaceModel::estimateStateSigma of the oracle against the NumPy restatement (ProjectiveBase.cc:201-213), at a non-identity state"""
    ssm = oracle.SSM(oracle.SSM_HOM, 14, 11)
    ssm.set_corners(G["ess_corners"]); ssm.set_state(G["ess_p"])
    np.testing.assert_allclose(ssm.estimate_state_sigma(1.3), G["ess_hom"], rtol=1e-10)
    ssa = oracle.SSM(oracle.SSM_AFF, 14, 11)
    ssa.set_corners(G["ess_corners"]); ssa.set_state(G["ess_pa"])
    np.testing.assert_allclose(ssa.estimate_state_sigma(1.3), G["ess_aff"], rtol=1e-10)


def test_golden_generator_is_reproducible(tmp_path):
    """The committed fixture is exactly what the committed generator produces."""
    import subprocess, sys, shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = os.path.join(root, "tests", "golden", "make_golden.py")
    keep = os.path.join(root, "tests", "golden", "lk_golden.npz")
    backup = tmp_path / "orig.npz"
    shutil.copy(keep, backup)
    try:
        subprocess.check_call([sys.executable, gen], stdout=subprocess.DEVNULL)
        new = np.load(keep)
        old = np.load(backup)
        assert sorted(new.files) == sorted(old.files)
        for k in new.files:
            np.testing.assert_array_equal(new[k], old[k])
    finally:
        shutil.copy(backup, keep)
