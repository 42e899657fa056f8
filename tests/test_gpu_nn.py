"""r06: nt::NN's dataset generation on the device (SM/src/NT/NN.cc:131-191) as one launch -- mtfhip_nn_dataset / _dev, mtf_amd.sm.NNDataset --
against the oracle's restatement of generateDataset (invertState -> compositionalUpdate -> updatePixVals -> updateDistFeat ->
compositionalUpdate, the reference's walk with its rounding) for the SSD, NCC and MI distance features, both state-space models, one and
three channels; the device's own perturbation draws (statistics, determinism, the row-block property the sharded form rests on); the full
size of the bench workload through size-independent properties; refusals."""
import numpy as np
import pytest

import mtf_amd
from mtf_amd import _lib as L
from mtf_amd import synth
from mtf_amd.sm import NNDataset

pytestmark = pytest.mark.gpu

SIGMA_H = np.array([0.02, 0.02, 2.0, 0.02, 0.02, 2.0, 1e-4, 1e-4])
SIGMA_A = np.array([2.0, 2.0, 0.02, 0.02, 0.02, 0.02])


def _oracle_pair(oracle, am, ssm, res, img, corners, channels=1, **am_kw):
    o_ssm = oracle.SSM(ssm, res, res); o_am = oracle.AM(am, res, res, **am_kw)
    if channels > 1:
        o_am.set_channels(channels); o_ssm.set_channels(channels)
    o_am.set_curr_img(img)
    o_ssm.set_corners(corners)
    o_am.initialize_pix_vals(o_ssm.get("curr_pts"))      # NN::initialize NT/NN.cc:85-90
    return o_am, o_ssm


@pytest.mark.parametrize("ssm", [L.SSM_HOMOGRAPHY, L.SSM_AFFINE], ids=["hom", "aff"])
@pytest.mark.parametrize("am,am_kw", [(L.AM_SSD, {}), (L.AM_NCC, {}), (L.AM_MI, {}), (L.AM_MI, dict(mi_n_bins=10, mi_pou=1))], ids=["ssd", "ncc", "mi8", "mi10pou"])
def test_nn_dataset_rows_follow_oracle(oracle, gpu_ctx, frame, parity_record, am, am_kw, ssm):
    rng = np.random.default_rng(71)
    res, n = 24, 48
    corners = synth.square_corners(250, 262, 90) + rng.uniform(-2, 2, size=(2, 4))
    S = 8 if ssm == L.SSM_HOMOGRAPHY else 6
    perts = rng.normal(size=(n, S)) * (SIGMA_H if S == 8 else SIGMA_A)
    perts[0] = 0
    o_kw = {{"mi_pou": "pou", "mi_n_bins": "n_bins"}.get(k, k): v for k, v in am_kw.items()}
    o_am, o_ssm = _oracle_pair(oracle, am, ssm, res, frame, corners, **o_kw)
    want = oracle.nn_generate_dataset(o_am, o_ssm, perts)
    gpu_ctx.set_image(frame)
    ds = NNDataset(gpu_ctx, am=am, ssm=ssm, resx=res, resy=res, n_samples=n, am_params=am_kw)
    got = ds.initialize(corners, perts)
    N = res * res
    assert got.shape == want.shape == (n, 5 * N if am == L.AM_MI else N) and ds.feature_size() == got.shape[1]
    np.testing.assert_array_equal(ds.perturbations, perts)
    if am == L.AM_MI:
        # row 0 of the 5 x N matrix is floor(It): the same integers (a pixel value within 1e-9 of an integer could land on either side: none here)
        assert np.array_equal(got[:, :N], want[:, :N])
        np.testing.assert_allclose(got[:, N:], want[:, N:], rtol=0, atol=1e-9)
        assert np.allclose(got[:, N:].reshape(n, 4, N).sum(axis=1)[(got[:, :N] >= 1) & (got[:, :N] <= (am_kw.get("mi_n_bins", 8) - 3))], 1.0, atol=1e-9)   # interior windows sum to one
    else:
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 if am == L.AM_SSD else 1e-12)
    parity_record.append(dict(test="nn_dataset_rows", am=int(am), ssm=int(ssm), **{k: int(v) for k, v in am_kw.items()}, max_abs=float(np.abs(got - want).max())))
    # the zero perturbation: the template's own feature, and the SSM is where it was
    np.testing.assert_allclose(ds.batch.get_state()[0], 0.0, atol=0)
    ds.batch.close()


@pytest.mark.parametrize("am", [L.AM_SSD, L.AM_NCC, L.AM_MI], ids=["mcssd", "mcncc", "mcmi"])
def test_nn_dataset_multichannel_rows_follow_oracle(oracle, gpu_ctx, am):
    rng = np.random.default_rng(72)
    res, n = 20, 24
    img = synth.make_frame_mc(256, 256)
    corners = synth.square_corners(128.0, 120.0, 70.0) + rng.uniform(-1, 1, size=(2, 4))
    perts = rng.normal(size=(n, 8)) * SIGMA_H
    o_am, o_ssm = _oracle_pair(oracle, am, L.SSM_HOMOGRAPHY, res, img, corners, channels=3)
    want = oracle.nn_generate_dataset(o_am, o_ssm, perts)
    gpu_ctx.set_image(img)
    ds = NNDataset(gpu_ctx, am=am, ssm=L.SSM_HOMOGRAPHY, resx=res, resy=res, n_samples=n, am_params=dict(n_channels=3))
    got = ds.initialize(corners, perts)
    N = 3 * res * res
    assert got.shape == want.shape == (n, 5 * N if am == L.AM_MI else N)
    if am == L.AM_MI:
        assert np.array_equal(got[:, :N], want[:, :N])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 if am != L.AM_NCC else 1e-12)
    ds.batch.close()


def test_nn_dataset_device_draws_and_row_blocks(gpu_ctx, frame):
    """the perturbations drawn on the device: N(mean_k, sigma_k) per component (ProjectiveBase.cc:283-288), a pure function of (seed, global
    sample index) -- so any partition of the rows into blocks reproduces the unpartitioned matrix BIT FOR BIT (what the sharded form's
    all-gather rests on) -- and the rows are the features of exactly those perturbations"""
    import torch
    gpu_ctx.set_image(frame)
    res, n = 20, 20000
    corners = synth.square_corners(250, 262, 80)
    mean = np.array([0.001, 0, 0.5, 0, -0.001, -0.25, 0, 0])
    b = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, res, res, 1)
    b.set_corners(corners[None]); b.initialize_pix_vals()
    p1, f1 = b.nn_dataset(n, SIGMA_H, mean, seed=11)
    p2, f2 = b.nn_dataset(n, SIGMA_H, mean, seed=11)
    p3, _ = b.nn_dataset(n, SIGMA_H, mean, seed=12)
    assert np.array_equal(p1, p2) and np.array_equal(f1, f2) and not np.array_equal(p1, p3)
    z = (p1 - mean) / SIGMA_H
    assert np.abs(z.mean(axis=0)).max() < 4.0 / np.sqrt(n) and np.abs(z.std(axis=0) - 1).max() < 0.03
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03 and abs(np.corrcoef(z[:-1, 2], z[1:, 2])[0, 1]) < 0.03   # the two halves of a Box-Muller pair; neighbouring samples
    assert (np.abs(z) > 4).sum() < 20      # ~ 6e-5 of 160 000 draws
    # given back as the caller's perturbations they reproduce the rows
    _, f4 = b.nn_dataset(256, SIGMA_H, mean, seed=0, perturbations=p1[:256])
    assert np.array_equal(f4, f1[:256])
    # row blocks through the device form, in a different order and with ragged sizes
    d = b.nn_desc(n, SIGMA_H, mean, 11)
    buf = torch.zeros((n, res * res), dtype=torch.float64, device="cuda:0")
    for lo, cnt in ((15000, 5000), (0, 7001), (7001, 7999)):
        b.nn_dataset_dev(d, buf[lo:].data_ptr(), lo, cnt)
    gpu_ctx.synchronize()
    assert np.array_equal(buf.cpu().numpy(), f1)
    # zero sigma: every sample is the template -- to rounding in tolerance mode (k_nn_rows: reciprocal + factored interpolant), bit
    # for bit in the reference's operation order
    _, f0 = b.nn_dataset(16, np.zeros(8), None, seed=3)
    np.testing.assert_allclose(f0, np.repeat(b.read(L.BUF_I0), 16, axis=0), rtol=0, atol=1e-11)
    b.set_math_mode(mtf_amd.MATH_REPLAY)
    _, f0 = b.nn_dataset(16, np.zeros(8), None, seed=3)
    assert np.array_equal(f0, np.repeat(b.read(L.BUF_I0), 16, axis=0))
    b.close()


@pytest.mark.parametrize("ssm", [L.SSM_HOMOGRAPHY, L.SSM_AFFINE], ids=["hom", "aff"])
@pytest.mark.parametrize("am,am_kw", [(L.AM_SSD, {}), (L.AM_NCC, {}), (L.AM_MI, dict(mi_n_bins=10, mi_pou=1))], ids=["ssd", "ncc", "mi10pou"])
@pytest.mark.parametrize("where", ["inside", "border"])
def test_nn_two_launch_form_equals_workgroup_form(gpu_ctx, frame, am, am_kw, ssm, where):
    """tolerance mode's two launches (k_nn_warps: a thread per sample forms its warp; k_nn_rows: a workgroup per sample, a quarter row per wave)
    against the workgroup-per-sample form in the reference's operation order (MATH_REPLAY): the same perturbations bit for bit (one sampler), rows to 1e-9 of a pixel value -- for templates whose samples
    stay inside the frame (no border test at all) and for ones that cross it (border value 128, mixed waves)"""
    gpu_ctx.set_image(frame)
    res, n = 50, 333                   # (2500 entries: 19.5 pair-rounds of 128 -- a ragged last round and an idle tail in the last wave; 333 samples: a ragged last block of k_nn_warps)
    h, w = frame.shape[:2]
    corners = synth.square_corners(w / 2, h / 2, 100) if where == "inside" else synth.square_corners(w - 40.0, 35.0, 100)
    S = 8 if ssm == L.SSM_HOMOGRAPHY else 6
    sigma = (SIGMA_H if S == 8 else SIGMA_A) * 2.0
    rows = {}
    for mode in (mtf_amd.MATH_FAST, mtf_amd.MATH_REPLAY):
        b = mtf_amd.Batch(gpu_ctx, am, ssm, res, res, 1, **am_kw)
        b.set_math_mode(mode)
        b.set_corners(corners[None]); b.initialize_pix_vals()
        rows[mode] = b.nn_dataset(n, sigma, None, seed=5)
        b.close()
    (pf, ff), (pr, fr) = rows[mtf_amd.MATH_FAST], rows[mtf_amd.MATH_REPLAY]
    assert np.array_equal(pf, pr)
    N = res * res
    if where == "border" and am == L.AM_SSD:
        assert (fr == 128.0).mean() > 0.05          # samples did leave the frame (border value 128)
    if am == L.AM_MI:
        keep = (ff[:, :N] == fr[:, :N])             # floor(It): a value within rounding of an integer may land on either side
        assert (~keep).mean() < 1e-6
        d = np.abs(ff[:, N:].reshape(n, 4, N) - fr[:, N:].reshape(n, 4, N)).max(axis=1)
        assert d[keep].max() < 1e-9
    else:
        np.testing.assert_allclose(ff, fr, rtol=0, atol=1e-9 if am == L.AM_SSD else 1e-12)


def test_nn_dataset_full_size_properties_and_refusals(gpu_ctx, frame):
    """the bench workload's shape (10 000 samples of 50 x 50, BASELINE's config-4 candidate set as an NN dataset) through properties: NCC rows
    are zero-mean and of unit norm; the nearest row of a stored sample is itself; the sampler's spread shows in the features' distance to
    the template; several distributions = consecutive row blocks"""
    gpu_ctx.set_image(frame)
    corners = synth.square_corners(256, 256, 100)
    ds = NNDataset(gpu_ctx, am=L.AM_NCC, resx=50, resy=50, n_samples=10000, ssm_sigma=(SIGMA_H * 0.1, SIGMA_H), distr_n_samples=[2500, 7500], seed=9)
    f = ds.initialize(corners)
    assert f.shape == (10000, 2500) and ds.perturbations.shape == (10000, 8)
    np.testing.assert_allclose(np.linalg.norm(f, axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(f.sum(axis=1), 0.0, atol=1e-10)
    k, dist = ds.nearest(f[4321])
    assert k == 4321 and dist == 0.0
    ds0 = NNDataset(gpu_ctx, am=L.AM_NCC, resx=50, resy=50, n_samples=1, ssm_sigma=np.zeros(8))
    t = ds0.initialize(corners)[0]
    d_small, d_large = np.linalg.norm(f[:2500] - t, axis=1).mean(), np.linalg.norm(f[2500:] - t, axis=1).mean()
    assert d_small < 0.5 * d_large
    assert np.abs(ds.perturbations[:2500, 2]).std() < 0.5 * np.abs(ds.perturbations[2500:, 2]).std()
    # refusals
    b = ds.batch
    d = b.nn_desc(100, SIGMA_H)
    d.additive_update = 1
    with pytest.raises(mtf_amd.FunctionNotImplemented):
        b.nn_dataset_dev(d, 1, 0, 1)
    d.additive_update = 0
    with pytest.raises(mtf_amd.MtfHipError, match="rows"):
        b.nn_dataset_dev(d, 1, 90, 20)
    b2 = mtf_amd.Batch(gpu_ctx, L.AM_SSD, L.SSM_HOMOGRAPHY, 10, 10, 2)
    b2.set_corners(np.stack([corners, corners]))
    with pytest.raises(mtf_amd.MtfHipError, match="one template"):
        b2.nn_dataset(4, SIGMA_H)
    b2.close(); ds.batch.close(); ds0.batch.close()
