"""The HIP path against the r04 fixtures directly (tests/golden/lk_golden2.npz: oracle/numpy_ref.py via tests/golden/make_golden2.py) --
the cases tests/test_oracle_golden2.py pins the C++ oracle with, without the oracle in between: NCC and MI second-order Hessians
(AM/src/NCC.cc:391-410, AM/src/MI.cc:659-735), particle-filter resampling and estimates (SM/src/NT/PF.cc:345-614), multi-channel
sampling and gradients (Utilities/src/imgUtils.cc:861-1005)."""
import os

import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import synth
from mtf_amd.sm import ParticleFilter

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden2.npz"))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture
def gimg(gpu_ctx):
    img = synth.make_frame(*[int(v) for v in G["img_shape"]], seed=int(G["img_seed"]))
    gpu_ctx.set_image(img)
    return img


def _second_order_batch(gpu_ctx, am):
    res = 22
    b = mtf_amd.Batch(gpu_ctx, am, L.SSM_AFFINE, res, res, 1)
    b.set_corners(G["so2_corners"][None])
    b.initialize_pix_vals(); b.initialize_pix_grad(); b.initialize_pix_hess()
    b.initialize_similarity(); b.initialize_grad(); b.initialize_hess()
    b.cmpt_pix_jacobian(L.JAC_WARPED, L.BUF_DI0_DX, L.BUF_J0)                      # at the identity warp
    b.cmpt_pix_hessian(L.JAC_WARPED, L.BUF_D2I0_DX2, L.BUF_DI0_DX, L.BUF_D2I0_DP2)
    b.set_state(G["so2_p"][None])
    b.update_pix_vals(); b.update_pix_grad(); b.update_pix_hess()
    b.update_similarity(False); b.update_curr_grad(); b.update_init_grad()
    b.cmpt_warped_pix_jacobian()
    b.cmpt_pix_hessian(L.JAC_WARPED, L.BUF_D2IT_DX2, L.BUF_DIT_DX, L.BUF_D2IT_DP2)
    return b


def test_ncc_second_order_hessians_golden(gpu_ctx, gimg):
    b = _second_order_batch(gpu_ctx, L.AM_NCC)
    assert rel(b.cmpt_curr_hessian()[0], G["ncc_H_curr1"]) < 1e-5
    assert rel(b.cmpt_init_hessian()[0], G["ncc_H_init1"]) < 1e-5
    assert rel(b.cmpt_curr_hessian2()[0], G["ncc_H_curr2"]) < 1e-5
    assert rel(b.cmpt_init_hessian2()[0], G["ncc_H_init2"]) < 1e-5
    b.close()


def test_mi_second_order_hessians_golden(gpu_ctx, gimg):
    b = _second_order_batch(gpu_ctx, L.AM_MI)
    assert rel(b.cmpt_self_hessian()[0], G["mi_H_self1"]) < 1e-5
    assert rel(b.cmpt_curr_hessian2()[0], G["mi_H_curr2"]) < 1e-5
    assert rel(b.cmpt_self_hessian2()[0], G["mi_H_self2"]) < 1e-5
    b.close()


@pytest.mark.parametrize("math", [mtf_amd.MATH_REPLAY, mtf_amd.MATH_FAST])
@pytest.mark.parametrize("mean_type,resampling_type", [(1, 1), (2, 2), (0, 1), (1, 3)])
def test_pf_resampling_and_estimates_golden(gpu_ctx, gimg, mean_type, resampling_type, math):
    """one iteration of the device filter with zero draws (the proposals are the given states): the weights, the resampled set and the
    estimate of every mean type against the NumPy restatement"""
    n, res = 48, 20
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, res, res, n_particles=n, ssm_sigma=(1.0,) * 8, corner_based_sampling=0, dynamic_model=0,
                        update_type=1, likelihood_func=0, resampling_type=resampling_type, mean_type=mean_type,
                        likelihood_alpha=float(G["pf2_alpha"]), seed=9)
    pf.batch.set_math_mode(math)
    pf.initialize(G["pf2_corners"][None])
    pf.set_particles(G["pf2_states"], np.zeros((n, 8)))
    pf.iteration(np.zeros((n, 8)), G["pf2_uniforms"])
    st, ar, w, ids = pf.particles()
    np.testing.assert_allclose(w, G["pf2_wts"] / (G["pf2_wts"].sum() if resampling_type == 3 else 1.0), rtol=1e-9)
    want = G["pf2_ids_residual"] if resampling_type == 3 else G["pf2_ids_multinomial"]
    np.testing.assert_allclose(st, G["pf2_states"][want], rtol=0, atol=1e-14)
    if resampling_type != 3:
        assert np.array_equal(ids, want)
    if mean_type == 1 and resampling_type == 1:
        np.testing.assert_allclose(pf.batch.get_state()[0], G["pf2_mean_state"], rtol=1e-11, atol=1e-14)
    if mean_type == 2:
        np.testing.assert_allclose(pf.get_region()[0], G["pf2_mean_corners"], rtol=0, atol=1e-9)
    if mean_type == 0:
        np.testing.assert_allclose(pf.batch.get_state()[0], st[int(G["pf2_max_wt_id_new_set"])], rtol=0, atol=1e-14)
    pf.close()


def test_mc_sampling_golden(gpu_ctx):
    img3 = synth.make_frame_mc(96, 96, seed=int(G["mc_img_seed"]))
    gpu_ctx.set_image(img3)
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 12, 10, 1, n_channels=3)
    b.set_corners(G["mc_corners"][None])
    b.initialize_pix_vals(); b.initialize_pix_grad()
    np.testing.assert_allclose(b.read(L.BUF_I0)[0], G["mc_I0"], rtol=0, atol=1e-10)
    b.set_state(G["mc_p"][None])
    b.update_pix_vals(); b.update_pix_grad()
    np.testing.assert_allclose(b.read(L.BUF_IT)[0], G["mc_It"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(b.read(L.BUF_DIT_DX)[0], G["mc_dIt_dx"], rtol=0, atol=5e-5)
    b.close()
