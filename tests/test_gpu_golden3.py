"""The HIP path against the r05 fixtures directly (tests/golden/lk_golden3.npz: oracle/numpy_ref.py via tests/golden/make_golden3.py),
without the oracle in between: GridTracker's patch layout as the device trackers receive it (SM/src/GridTracker.cc:345-380), NN
dataset rows (SM/src/NT/NN.cc:131-191), the device filter's samplers for given normals (Homography.cc:899-909, Affine.cc:464-494)."""
import os

import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import synth
from mtf_amd.sm import GridTracker, NNDataset, ParticleFilter

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden3.npz"))
MODES = {"inside": (0, 1), "points": (0, 0), "dyn": (1, 0)}


@pytest.fixture
def gimg(gpu_ctx):
    img = synth.make_frame(*[int(v) for v in G["img_shape"]], seed=int(G["img_seed"]))
    gpu_ctx.set_image(img)
    return img


@pytest.mark.parametrize("mode", sorted(MODES))
def test_grid_layout_golden(gpu_ctx, gimg, mode):
    """the patches GridTracker.initialize hands the device's patch trackers (read back from the batch), the grid SSM's points, and
    the centroids it keeps as prev_pts (rounded to float like cv::Point2f)"""
    gx, gy, px, py = [int(v) for v in G["grid_dims"]]
    dyn, inside = MODES[mode]
    g = GridTracker(gpu_ctx, grid_size=gx, grid_size_y=gy, patch_size=px, patch_size_y=py, am=L.AM_SSD, ssm=L.SSM_HOMOGRAPHY, max_iters=5,
                    dyn_patch_size=dyn, patch_centroid_inside=inside)
    assert g.res()[0] * g.res()[1] == len(G["grid_pts_" + mode])
    np.testing.assert_allclose(g.grid_pts(G["grid_region"]), G["grid_pts_" + mode], rtol=0, atol=1e-9)
    g.initialize(G["grid_region"])
    np.testing.assert_allclose(g.tracker.get_region(), G["grid_patches_" + mode], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(g.prev_pts, (G["grid_patches_" + mode].sum(axis=2) / 4.0).astype(np.float32))
    # a frame on the same image leaves every patch where the layout put it
    g.update_patches()
    np.testing.assert_allclose(g.tracker.get_region(), G["grid_patches_" + mode], rtol=0, atol=1e-5)
    g.tracker.batch.close()


@pytest.mark.parametrize("am", ["ssd", "ncc"])
def test_nn_dataset_rows_golden(gpu_ctx, gimg, am):
    ds = NNDataset(gpu_ctx, am=L.AM_SSD if am == "ssd" else L.AM_NCC, resx=24, resy=24, n_samples=len(G["nn_perts"]))
    feats = ds.initialize(G["nn_corners"], G["nn_perts"])
    np.testing.assert_allclose(feats, G["nn_rows_" + am], rtol=0, atol=1e-8 if am == "ssd" else 1e-11)
    ds.batch.close()


G4 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lk_golden4.npz"))


@pytest.mark.parametrize("math", [mtf_amd.MATH_REPLAY, mtf_amd.MATH_FAST], ids=["replay", "fast"])
@pytest.mark.parametrize("n_bins,pou", [(8, 0), (10, 1)], ids=["8", "10pou"])
def test_nn_dataset_mi_rows_golden(gpu_ctx, gimg, n_bins, pou, math):
    """nt::NN's dataset for the MI appearance model (MI.cc:80-94, 736-747) in both arithmetic modes (workgroup-per-sample kernel / k_nn_warps +
    k_nn_rows) against lk_golden4's NumPy rows: floor(It) exactly (none of these values sits within rounding of an integer), weights to 1e-8"""
    ds = NNDataset(gpu_ctx, am=L.AM_MI, resx=24, resy=24, n_samples=len(G["nn_perts"]), am_params=dict(mi_n_bins=n_bins, mi_pou=pou))
    ds.batch.set_math_mode(math)
    feats = ds.initialize(G["nn_corners"], G["nn_perts"])
    want = G4["nn_mi_rows_%s" % ("10pou" if pou else "8")]
    N = 24 * 24
    assert feats.shape == want.shape
    assert np.array_equal(feats[:, :N], want[:, :N])
    np.testing.assert_allclose(feats[:, N:], want[:, N:], rtol=0, atol=1e-8)
    ds.batch.close()


@pytest.mark.parametrize("math", [mtf_amd.MATH_REPLAY, mtf_amd.MATH_FAST])
def test_homography_corner_sampler_golden(gpu_ctx, gimg, math):
    """one iteration of the device filter without resampling, RandomWalk + compositional from the identity: the particles are the
    corner based perturbations of the given normals"""
    n = len(G["smp_hom_z"])
    pf = ParticleFilter(gpu_ctx, L.SSM_HOMOGRAPHY, 16, 16, n_particles=n, ssm_sigma=tuple(G["smp_hom_sigma"]), ssm_mean=tuple(G["smp_hom_mean"]),
                        corner_based_sampling=1, dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=0, mean_type=0, seed=3)
    pf.batch.set_math_mode(math)
    pf.initialize(G["smp_hom_corners"][None])
    pf.iteration(G["smp_hom_z"], np.full(n, 0.5))
    st = pf.particles()[0]
    np.testing.assert_allclose(st, G["smp_hom_states"], rtol=1e-8, atol=1e-11)
    pf.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_affine_point_sampler_golden(gpu_ctx, gimg, mode):
    n = len(G["smp_aff_z"])
    pf = ParticleFilter(gpu_ctx, L.SSM_AFFINE, 16, 16, n_particles=n, ssm_sigma=tuple(G["smp_aff_sigma"]), ssm_mean=tuple(G["smp_aff_mean"]),
                        dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=0, mean_type=0, pt_based_sampling=mode, seed=3)
    pf.initialize(G["smp_aff_corners"][None])
    z = np.ascontiguousarray(G["smp_aff_z"][:, :8 if mode == 2 else 6])
    pf.iteration(z, np.full(n, 0.5))
    st = pf.particles()[0]
    np.testing.assert_allclose(st[:, :6], G["smp_aff_states_%d" % mode], rtol=1e-8, atol=1e-11)
    pf.close()
