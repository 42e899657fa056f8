"""The HIP path against the committed golden fixtures directly (tests/golden/lk_golden.npz: the NumPy re-derivation of
oracle/numpy_ref.py, generator tests/golden/make_golden.py) -- the same cases tests/test_oracle_golden.py pins the C++ oracle
with, now without the oracle in between.  Tolerances are the fixtures': bit-level quantities 1e-9, anything that contains the
reference's 1e-8 finite-difference gradient 1e-5 (its own quantisation noise)."""
import os

import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden.npz"))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture
def gimg(gpu_ctx):
    img = synth.make_frame(*[int(v) for v in G["img_shape"]], seed=int(G["img_seed"]))
    gpu_ctx.set_image(img)
    return img


@pytest.mark.parametrize("math", [mtf_amd.MATH_REPLAY, mtf_amd.MATH_FAST])
@pytest.mark.parametrize("tag", ["sq", "quad"])
def test_homography_ssd_step_golden(gpu_ctx, gimg, tag, math):
    res = int(G[tag + "_res"])
    corners, p = G[tag + "_corners"], G[tag + "_p"]
    for sm_kind, key, mat in ((L.SM_FCLK, "fclk", 1), (L.SM_ESM, "esm", 1), (L.SM_FCLK, "fclk", 0), (L.SM_ESM, "esm", 0)):
        b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, res, res, 1)
        b.set_math_mode(math)
        sm = mtf_amd.sm_desc(sm_kind, materialize=mat, leven_marq=0)
        b.set_corners(corners[None])
        np.testing.assert_allclose(b.read(L.BUF_INIT_PTS)[0], G[tag + "_init_pts_full"], atol=1e-9)
        b.init_template(sm)
        np.testing.assert_allclose(b.read(L.BUF_I0)[0], G[tag + "_I0_full"], atol=1e-9)
        assert rel(b.read(L.BUF_J0)[0], G[tag + "_J0_full"]) < 1e-5
        b.set_state(p[None])
        f, g, H = b.iterate(sm)
        assert abs(f[0] - float(G[tag + "_f"])) <= 1e-10 * abs(float(G[tag + "_f"]))
        assert rel(H[0], G[tag + "_%s_H" % key]) < 1e-5
        assert rel(g[0], G[tag + "_%s_g" % key]) < 1e-5
        assert rel(-np.linalg.solve(H[0], g[0]), G[tag + "_%s_dp" % key]) < 1e-5
        if mat:     # the interface-visible arrays of the materialising launch
            np.testing.assert_allclose(b.read(L.BUF_IT)[0], G[tag + "_It_full"], atol=1e-9)
            np.testing.assert_allclose(b.read(L.BUF_DIT_DX)[0], G[tag + "_grad_full"], atol=5e-5)
            if sm_kind == L.SM_FCLK:
                assert rel(b.read(L.BUF_JT)[0], G[tag + "_Jt_full"]) < 1e-5
        b.close()


def test_affine_ncc_golden(gpu_ctx, gimg):
    res = 25
    b = mtf_amd.Batch(gpu_ctx, L.AM_NCC, L.SSM_AFFINE, res, res, 1)
    b.set_corners(G["ncc_corners"][None])
    b.initialize_pix_vals(); b.initialize_pix_grad(); b.initialize_similarity(); b.initialize_grad(); b.initialize_hess()
    b.cmpt_pix_jacobian(L.JAC_WARPED, L.BUF_DI0_DX, L.BUF_J0)
    assert rel(b.read(L.BUF_J0)[0][:16], G["ncc_J0_head"]) < 1e-5
    b.set_state(G["ncc_p"][None])
    b.update_pix_vals(); b.update_similarity(False); b.update_init_grad(); b.update_curr_grad()
    assert abs(b.get_similarity()[0] - float(G["ncc_f"])) < 1e-12
    np.testing.assert_allclose(b.read(L.BUF_DF_DIT)[0][:16], G["ncc_df_dIt_head"], rtol=1e-9, atol=1e-15)
    np.testing.assert_allclose(b.read(L.BUF_DF_DI0)[0][:16], G["ncc_df_dI0_head"], rtol=1e-9, atol=1e-15)
    assert rel(b.cmpt_init_jacobian()[0], G["ncc_g_init"]) < 1e-5
    assert rel(b.cmpt_self_hessian(L.BUF_J0)[0], G["ncc_H_self_J0"]) < 1e-5
    b.close()


def test_mi_golden(gpu_ctx, gimg):
    """MI (8 bins, 40 x 40 homography patch): f, df_dIt . Jt and cmptCurrHessian(Jt) of the device kernels against the NumPy
    evaluation of the definition."""
    res = 40
    b = mtf_amd.Batch(gpu_ctx, L.AM_MI, L.SSM_HOMOGRAPHY, res, res, 1)
    b.set_corners(G["mi_corners"][None])
    b.initialize_pix_vals(); b.initialize_pix_grad(); b.initialize_similarity(); b.initialize_grad(); b.initialize_hess()
    b.set_state(G["mi_p"][None])
    b.update_pix_vals(); b.update_similarity(False); b.update_curr_grad(); b.update_pix_grad()
    np.testing.assert_allclose(b.read(L.BUF_IT)[0][:16], G["mi_It_head"], atol=1e-9)
    assert abs(b.get_similarity()[0] - float(G["mi_f"])) <= 1e-10 * abs(float(G["mi_f"]))
    np.testing.assert_allclose(b.read(L.BUF_DF_DIT)[0][:16], G["mi_df_dIt_head"], rtol=1e-8, atol=1e-14)
    b.cmpt_warped_pix_jacobian()
    assert rel(b.cmpt_curr_jacobian()[0], G["mi_g_curr"]) < 1e-5
    assert rel(b.cmpt_curr_hessian()[0], G["mi_H_curr"]) < 1e-5
    b.close()


def test_second_order_affine_ssd_golden(gpu_ctx, gimg):
    res = 22
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_AFFINE, res, res, 1)
    b.set_corners(G["so_corners"][None])
    b.initialize_pix_vals(); b.initialize_pix_grad(); b.initialize_pix_hess()
    b.initialize_similarity(); b.initialize_grad(); b.initialize_hess()
    b.set_state(G["so_p"][None])
    b.update_pix_vals(); b.update_pix_grad(); b.update_pix_hess()
    b.update_similarity(False); b.update_curr_grad(); b.update_init_grad()
    np.testing.assert_allclose(b.read(L.BUF_D2IT_DX2)[0].reshape(-1, 4)[:16], G["so_img_hess_head"], rtol=0, atol=1e-10)
    b.cmpt_warped_pix_jacobian()
    b.cmpt_pix_hessian(L.JAC_WARPED, L.BUF_D2IT_DX2, L.BUF_DIT_DX, L.BUF_D2IT_DP2)
    assert rel(b.cmpt_curr_hessian2()[0], G["so_H_curr2"]) < 1e-5
    b.close()


@pytest.mark.parametrize("math", [mtf_amd.MATH_REPLAY, mtf_amd.MATH_FAST])
def test_pf_scores_golden(gpu_ctx, gimg, math):
    res = 20
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, res, res, 1, likelihood_alpha=1.0)
    b.set_math_mode(math)
    b.set_corners(G["pf_corners"][None])
    b.initialize_pix_vals(); b.initialize_similarity()
    lik = b.score_candidates(G["pf_states"])
    np.testing.assert_allclose(lik, G["pf_likelihood"], rtol=1e-9)
    b.close()


def test_estimate_state_sigma_golden(gpu_ctx, gimg):
    """mtfhip_ssm_estimate_state_sigma (StateSpaceModel::estimateStateSigma) against the fixture, at a non-identity state"""
    for ssm, key, pkey in ((L.SSM_HOMOGRAPHY, "ess_hom", "ess_p"), (L.SSM_AFFINE, "ess_aff", "ess_pa")):
        b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, ssm, 14, 11, 1)
        b.set_corners(G["ess_corners"][None])
        b.set_state(G[pkey][None])
        np.testing.assert_allclose(b.estimate_state_sigma(1.3)[0], G[key], rtol=1e-9)
        b.close()
