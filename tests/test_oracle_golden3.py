"""The C++ oracle -- and the host-only part of the C ABI -- against tests/golden/lk_golden3.npz (independent NumPy re-derivation,
tests/golden/make_golden3.py): GridTracker's patch layout, NN dataset rows with the reference's perturb -> sample -> un-perturb
sequence, the stochastic samplers for given draws."""
import ctypes as C
import os

import numpy as np
import pytest

from mtf_amd import _lib as L
from mtf_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden3.npz"))
MODES = {"inside": (0, 1), "points": (0, 0), "dyn": (1, 0)}


@pytest.fixture(scope="module")
def img():
    return synth.make_frame(*[int(v) for v in G["img_shape"]], seed=int(G["img_seed"]))


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("grid_ssm", [0, 1], ids=["hom", "aff"])
def test_grid_layout_golden(oracle, mode, grid_ssm):
    """GridTracker::resetTrackers' patches (GridTracker.cc:345-380) over a grid SSM of either kind: the oracle's restatement and the C
    ABI's mtfhip_grid_layout, both to 1e-9 px of the NumPy fixture on a strongly projective region"""
    gx, gy, px, py = [int(v) for v in G["grid_dims"]]
    dyn, inside = MODES[mode]
    res = oracle.grid_res(oracle.GridParams(gx, gy, px, py, 1, dyn, inside))
    assert res[0] * res[1] == len(G["grid_pts_" + mode])
    ssm = oracle.SSM(grid_ssm, res[0], res[1])
    g = oracle.Grid(ssm, grid_size=gx, grid_size_y=gy, patch_size=px, patch_size_y=py, dyn_patch_size=dyn, patch_centroid_inside=inside)
    g.initialize(G["grid_region"])
    np.testing.assert_allclose(ssm.get("curr_pts").reshape(-1, 2), G["grid_pts_" + mode], rtol=0, atol=1e-9)
    np.testing.assert_allclose(g.patch_corners(), G["grid_patches_" + mode], rtol=0, atol=1e-9)
    # the product's host arithmetic (no device needed)
    lib = C.CDLL(L.LIB_PATH)
    gd = L.GridDesc(gx, gy, px, py, 1, dyn, inside)
    pts, pcs = np.empty((res[0] * res[1], 2)), np.empty((gx * gy, 4, 2))
    r = np.ascontiguousarray(G["grid_region"].T)
    assert lib.mtfhip_grid_layout(C.byref(gd), r.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), pcs.ctypes.data_as(C.c_void_p)) == 0
    np.testing.assert_allclose(pts, G["grid_pts_" + mode], rtol=0, atol=1e-9)
    np.testing.assert_allclose(pcs.transpose(0, 2, 1), G["grid_patches_" + mode], rtol=0, atol=1e-9)


@pytest.mark.parametrize("am", ["ssd", "ncc"])
def test_nn_dataset_rows_golden(oracle, img, am):
    """NN::generateDataset (NT/NN.cc:131-191): invertState -> compositionalUpdate -> updatePixVals -> updateDistFeat -> compositionalUpdate"""
    res = 24
    ssm = oracle.SSM(oracle.SSM_HOM, res, res)
    o_am = oracle.AM(oracle.AM_SSD if am == "ssd" else oracle.AM_NCC, res, res)
    o_am.set_curr_img(img)
    ssm.set_corners(G["nn_corners"])
    o_am.initialize_pix_vals(ssm.get("curr_pts"))   # NN::initialize (NT/NN.cc:85-90)
    for k, p in enumerate(G["nn_perts"]):
        ssm.compositional_update(ssm.invert_state(p))
        o_am.update_pix_vals(ssm.get("curr_pts"))
        It = o_am.get("It")
        feat = It if am == "ssd" else (It - It.mean()) / np.linalg.norm(It - It.mean())   # SSDBase.h:116-125 / NCC.cc:530-537
        np.testing.assert_allclose(feat, G["nn_rows_" + am][k], rtol=0, atol=1e-9 if am == "ssd" else 1e-12, err_msg="row %d" % k)
        ssm.compositional_update(p)


G4 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden4.npz"))


@pytest.mark.parametrize("n_bins,pou", [(8, 0), (10, 1)], ids=["8", "10pou"])
def test_nn_dataset_mi_rows_golden(oracle, img, n_bins, pou):
    """the same dataset for the MI appearance model: MI's pixel normalisation (MI.cc:80-94) + MI::updateDistFeat (MI.cc:736-747) -- the oracle's
    generateDataset against lk_golden4's rows (NumPy, from lk_golden3's raw rows): floor(It) exactly, the four B-spline weights to 1e-9"""
    res = 24
    ssm = oracle.SSM(oracle.SSM_HOM, res, res)
    o_am = oracle.AM(oracle.AM_MI, res, res, n_bins=n_bins, pou=pou)
    o_am.set_curr_img(img)
    ssm.set_corners(G["nn_corners"])
    o_am.initialize_pix_vals(ssm.get("curr_pts"))
    got = oracle.nn_generate_dataset(o_am, ssm, G["nn_perts"])
    want = G4["nn_mi_rows_%s" % ("10pou" if pou else "8")]
    N = res * res
    assert got.shape == want.shape == (len(G["nn_perts"]), 5 * N)
    assert np.array_equal(got[:, :N], want[:, :N])
    np.testing.assert_allclose(got[:, N:], want[:, N:], rtol=0, atol=1e-9)


def test_homography_corner_sampler_golden(oracle, img):
    """Homography::generatePerturbation, corner based (Homography.cc:899-909), through one filter iteration without resampling: with
    RandomWalk + compositional updates from the identity the particles ARE the perturbations"""
    res, n = 16, len(G["smp_hom_z"])
    ssm = oracle.SSM(oracle.SSM_HOM, res, res); am = oracle.AM(oracle.AM_SSD, res, res); am.set_curr_img(img)
    ssm.set_corners(G["smp_hom_corners"])
    am.initialize_pix_vals(ssm.get("curr_pts")); am.initialize_similarity()
    pp = oracle.pf_params(n, dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=0, mean_type=0, corner_based_sampling=1,
                          sigma=tuple(G["smp_hom_sigma"]), mean=tuple(G["smp_hom_mean"]))
    st, ar, w, ids, mx = oracle.pf_iteration(am, ssm, pp, np.zeros((n, 8)), np.zeros((n, 8)), G["smp_hom_z"], np.full(n, 0.5), 0.0)
    np.testing.assert_allclose(st, G["smp_hom_states"], rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("mode", [1, 2])
def test_affine_point_sampler_golden(oracle, img, mode):
    """Affine::generatePerturbation with pt_based_sampling 1 / 2 (Affine.cc:464-494)"""
    res, n = 16, len(G["smp_aff_z"])
    ssm = oracle.SSM(oracle.SSM_AFF, res, res); am = oracle.AM(oracle.AM_SSD, res, res); am.set_curr_img(img)
    ssm.set_corners(G["smp_aff_corners"])
    am.initialize_pix_vals(ssm.get("curr_pts")); am.initialize_similarity()
    pp = oracle.pf_params(n, dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=0, mean_type=0, corner_based_sampling=0,
                          sigma=tuple(G["smp_aff_sigma"]), mean=tuple(G["smp_aff_mean"]), pt_based_sampling=mode)
    z = G["smp_aff_z"][:, :8 if mode == 2 else 6]
    st, ar, w, ids, mx = oracle.pf_iteration(am, ssm, pp, np.zeros((n, 6)), np.zeros((n, 6)), np.ascontiguousarray(z), np.full(n, 0.5), 0.0)
    np.testing.assert_allclose(st[:, :6], G["smp_aff_states_%d" % mode], rtol=1e-8, atol=1e-11)


def test_golden3_generator_is_reproducible(tmp_path):
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    keep = os.path.join(root, "tests", "golden", "lk_golden3.npz")
    backup = tmp_path / "orig.npz"
    shutil.copy(keep, backup)
    try:
        subprocess.check_call([sys.executable, os.path.join(root, "tests", "golden", "make_golden3.py")], stdout=subprocess.DEVNULL)
        new, old = np.load(keep), np.load(backup)
        assert sorted(new.files) == sorted(old.files)
        for k in new.files:
            np.testing.assert_array_equal(new[k], old[k])
    finally:
        shutil.copy(backup, keep)
