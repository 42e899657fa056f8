"""CPU: the C++ oracle against the r04 fixtures (tests/golden/lk_golden2.npz from the independent NumPy re-derivation via
tests/golden/make_golden2.py): NCC and MI second-order Hessians, particle-filter resampling and estimates, multi-channel sampling.
Tolerances as in test_oracle_golden.py: samples bit-level, anything downstream of the 1e-8 finite difference the north-star 1e-5."""
import os

import numpy as np
import pytest

from mtf_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden2.npz"))


@pytest.fixture(scope="module")
def img():
    return synth.make_frame(*[int(v) for v in G["img_shape"]], seed=int(G["img_seed"]))


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _second_order_state(oracle, img, am_kind):
    res = 22
    ssm = oracle.SSM(oracle.SSM_AFF, res, res)
    am = oracle.AM(am_kind, res, res)
    am.set_curr_img(img)
    ssm.set_corners(G["so2_corners"])
    pts0 = ssm.get("curr_pts")
    am.initialize_pix_vals(pts0); am.initialize_pix_grad_pts(pts0); am.initialize_pix_hess_pts(pts0)
    am.initialize_similarity(); am.initialize_grad(); am.initialize_hess()
    J0 = ssm.cmpt_warped_pix_jacobian(am.get("dI0_dx"))       # at the identity warp
    D0 = ssm.cmpt_warped_pix_hessian(am.get("d2I0_dx2"), am.get("dI0_dx"))
    ssm.set_state(G["so2_p"])
    pts = ssm.get("curr_pts")
    am.update_pix_vals(pts); am.update_pix_grad_pts(pts); am.update_pix_hess_pts(pts)
    am.update_similarity(False); am.update_curr_grad(); am.update_init_grad()
    Jt = ssm.cmpt_warped_pix_jacobian(am.get("dIt_dx"))
    Dt = ssm.cmpt_warped_pix_hessian(am.get("d2It_dx2"), am.get("dIt_dx"))
    return am, J0, D0, Jt, Dt


def test_ncc_second_order_hessians_golden(oracle, img):
    """NCC::cmptCurrHessian / cmptInitHessian, first order (NCC.cc:282-335, the reference's form incl. its `3` and its `/ b`) and with
    pixel Hessians (NCC.cc:391-410)"""
    am, J0, D0, Jt, Dt = _second_order_state(oracle, img, oracle.AM_NCC)
    assert rel(am.cmpt_curr_hessian(Jt), G["ncc_H_curr1"]) < 1e-5
    assert rel(am.cmpt_init_hessian(J0), G["ncc_H_init1"]) < 1e-5
    assert rel(am.cmpt_curr_hessian2(Jt, Dt), G["ncc_H_curr2"]) < 1e-5
    assert rel(am.cmpt_init_hessian2(J0, D0), G["ncc_H_init2"]) < 1e-5


def test_mi_second_order_hessians_golden(oracle, img):
    """MI::cmptCurrHessian and cmptSelfHessian with pixel Hessians (MI.cc:679-735) and the first-order self Hessian (MI.cc:515-601)"""
    am, J0, D0, Jt, Dt = _second_order_state(oracle, img, oracle.AM_MI)
    assert rel(am.cmpt_self_hessian(Jt), G["mi_H_self1"]) < 1e-5
    assert rel(am.cmpt_curr_hessian2(Jt, Dt), G["mi_H_curr2"]) < 1e-5
    assert rel(am.cmpt_self_hessian2(Jt, Dt), G["mi_H_self2"]) < 1e-5


@pytest.mark.parametrize("mean_type,resampling_type", [(1, 1), (2, 2), (0, 1), (1, 3)])
def test_pf_resampling_and_estimates_golden(oracle, img, mean_type, resampling_type):
    """one iteration of nt::PF::update's loop with zero draws (the proposals ARE the given states): weights, binary / linear multinomial
    and residual resampling, and the estimate of every mean type (NT/PF.cc:345-614) against the NumPy restatement"""
    n, res = 48, 20
    ssm = oracle.SSM(oracle.SSM_HOM, res, res)
    am = oracle.AM(oracle.AM_SSD, res, res, likelihood_alpha=float(G["pf2_alpha"]))
    am.set_curr_img(img)
    ssm.set_corners(G["pf2_corners"])
    am.initialize_pix_vals(ssm.get("curr_pts")); am.initialize_similarity()
    pp = oracle.pf_params(n, dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=resampling_type, mean_type=mean_type,
                          corner_based_sampling=0, sigma=(1.0,) * 8)
    st, ar, w, ids, mx = oracle.pf_iteration(am, ssm, pp, G["pf2_states"], np.zeros((n, 8)), np.zeros((n, 8)), G["pf2_uniforms"], 0.0)
    # (residualResampling normalises particle_wts in place, NT/PF.cc:539: the weights come back divided by their sum)
    np.testing.assert_allclose(w, G["pf2_wts"] / (G["pf2_wts"].sum() if resampling_type == 3 else 1.0), rtol=1e-9)
    if resampling_type in (1, 2):
        assert np.array_equal(ids, G["pf2_ids_multinomial"])
        np.testing.assert_allclose(st, G["pf2_states"][G["pf2_ids_multinomial"]], rtol=0, atol=1e-14)   # (a compositional update by the identity: 1 ulp)
    else:
        np.testing.assert_allclose(st, G["pf2_states"][G["pf2_ids_residual"]], rtol=0, atol=1e-14)
        assert mx == int(G["pf2_residual_best"])
    if mean_type == 1 and resampling_type == 1:
        np.testing.assert_allclose(ssm.get("state"), G["pf2_mean_state"], rtol=1e-11, atol=1e-14)
    if mean_type == 2:
        np.testing.assert_allclose(ssm.get("curr_corners").reshape(4, 2).T if ssm.get("curr_corners").shape != (2, 4) else ssm.get("curr_corners"),
                                   G["pf2_mean_corners"], rtol=0, atol=1e-9)
    if mean_type == 0:
        assert mx == int(G["pf2_max_wt_id_new_set"])
        np.testing.assert_array_equal(ssm.get("state"), st[mx])


def test_mc_sampling_golden(oracle):
    """mc::getPixVals / mc::getImgGrad (imgUtils.cc:861-1005): rows = (pixel, channel) pairs of a 32FC3 frame"""
    img3 = synth.make_frame_mc(96, 96, seed=int(G["mc_img_seed"]))
    ssm = oracle.SSM(oracle.SSM_HOM, 12, 10); ssm.set_channels(3)
    am = oracle.AM(oracle.AM_SSD, 12, 10); am.set_channels(3)
    am.set_curr_img(img3)
    ssm.set_corners(G["mc_corners"])
    am.initialize_pix_vals(ssm.get("curr_pts")); am.initialize_pix_grad_pts(ssm.get("curr_pts"))
    np.testing.assert_allclose(am.get("I0"), G["mc_I0"], rtol=0, atol=1e-10)
    ssm.set_state(G["mc_p"])
    pts = ssm.get("curr_pts")
    am.update_pix_vals(pts); am.update_pix_grad_pts(pts)
    np.testing.assert_allclose(am.get("It"), G["mc_It"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(am.get("dIt_dx").reshape(2, -1).T, G["mc_dIt_dx"], rtol=0, atol=5e-5)


def test_golden2_generator_is_reproducible(tmp_path):
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    keep = os.path.join(root, "tests", "golden", "lk_golden2.npz")
    backup = tmp_path / "orig.npz"
    shutil.copy(keep, backup)
    try:
        subprocess.check_call([sys.executable, os.path.join(root, "tests", "golden", "make_golden2.py")], stdout=subprocess.DEVNULL)
        new, old = np.load(keep), np.load(backup)
        assert sorted(new.files) == sorted(old.files)
        for k in new.files:
            np.testing.assert_array_equal(new[k], old[k])
    finally:
        shutil.copy(backup, keep)
