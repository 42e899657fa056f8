"""Pre-processing and pyramid levels.  CPU part: the NumPy restatement of OpenCV's float32 algorithms
against independent float64 mathematics.  GPU part: the device kernels against that restatement, bit for bit."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import preproc_ref as P  # noqa: E402

import mtf_amd  # noqa: E402
from mtf_amd import synth  # noqa: E402


def raw_frames():
    rng = np.random.default_rng(4)
    gray = rng.integers(0, 256, size=(97, 131), dtype=np.uint8)
    bgr = rng.integers(0, 256, size=(64, 83, 3), dtype=np.uint8)
    smooth = np.clip(synth.make_frame(160, 200), 0, 255).astype(np.uint8)
    return gray, bgr, smooth


def test_gaussian_kernel_and_blur_against_float64():
    k = P.gaussian_kernel5(3.0)
    x = np.arange(5) - 2.0
    ref = np.exp(-x * x / 18.0); ref /= ref.sum()
    np.testing.assert_allclose(k, ref, rtol=1e-6)
    assert k[0] == k[4] and k[1] == k[3] and abs(float(k.astype(np.float64).sum()) - 1) < 1e-6
    gray, _, _ = raw_frames()
    out = P.gaussian_blur5(gray.astype(np.float32))
    pad = np.pad(gray.astype(np.float64), 2, mode="reflect")           # numpy 'reflect' = BORDER_REFLECT_101
    acc = np.zeros(gray.shape)
    for dy in range(5):
        for dx in range(5):
            acc += ref[dy] * ref[dx] * pad[dy:dy + gray.shape[0], dx:dx + gray.shape[1]]
    np.testing.assert_allclose(out, acc, rtol=0, atol=2e-4)
    assert out.dtype == np.float32


def test_gray_conversion_and_pyramid_restatements():
    _, bgr, smooth = raw_frames()
    g = P.to_gray_f32(bgr)
    np.testing.assert_allclose(g, bgr @ np.array([0.114, 0.587, 0.299]), atol=1e-4)
    # pyrDown of a constant / of a linear ramp (interior): the 1-4-6-4-1 kernel preserves both
    const = np.full((40, 52), 7.25, dtype=np.float32)
    assert np.all(P.pyr_down(const, 20, 26) == const[:20, :26])
    yy, xx = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(80, dtype=np.float32), indexing="ij")
    ramp = (2 * xx + 3 * yy).astype(np.float32)
    d = P.pyr_down(ramp, 32, 40)
    np.testing.assert_allclose(d[2:-2, 2:-2], ramp[::2, ::2][2:-2, 2:-2], rtol=0, atol=1e-4)
    # INTER_LINEAR: identity when the size does not change, pixel-centre aligned when halving
    assert np.array_equal(P.resize_linear(ramp, 64, 80), ramp)
    h = P.resize_linear(ramp, 32, 40)
    np.testing.assert_allclose(h, 0.25 * (ramp[0::2, 0::2] + ramp[1::2, 0::2] + ramp[0::2, 1::2] + ramp[1::2, 1::2]), atol=1e-4)
    lvl = P.pyramid_level(smooth.astype(np.float32), 100, 125, use_pyr_down=False)
    assert lvl.shape == (100, 125) and lvl.dtype == np.float32


def test_equalize_hist_restatement_properties():
    """cv::equalizeHist as restated: a look-up table that is monotone, maps the lowest occupied level to 0 and the highest to
    255, flattens the cumulative distribution (every level's output ~ 255 x the fraction of pixels below it), leaves a constant
    image alone; convertTo(CV_8U) rounds to nearest even and saturates"""
    assert list(P.to_u8(np.array([-3.0, 0.5, 1.5, 2.5, 254.5, 255.5, 300.0], dtype=np.float32))) == [0, 0, 2, 2, 254, 255, 255]
    _, _, smooth = raw_frames()
    out = P.equalize_hist_u8(smooth)
    lo, hi = smooth.min(), smooth.max()
    assert out[smooth == lo].max() == 0 and out[smooth == hi].min() == 255
    order = np.argsort(smooth.ravel(), kind="stable")
    assert np.all(np.diff(out.ravel()[order].astype(int)) >= 0)            # monotone in the input level
    hist = np.bincount(smooth.ravel(), minlength=256)
    cdf = (np.cumsum(hist) - hist[lo]) / float(smooth.size - hist[lo])      # mass strictly above the lowest level, up to each level
    lev = np.nonzero(hist)[0]
    want = np.clip(np.rint(255.0 * cdf[lev]), 0, 255)
    got = np.array([out[smooth == v][0] for v in lev])
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1            # (float32 scale: at most one level off the float64 form)
    const = np.full((9, 11), 37, dtype=np.uint8)
    assert np.array_equal(P.equalize_hist_u8(const), const)
    full = P.preprocess(smooth, hist_eq=True, resize_factor=0.5)
    assert full.shape == (80, 100) and full.dtype == np.float32


@pytest.mark.gpu
def test_device_preprocess_hist_eq_and_resize(gpu_ctx):
    """hist_eq and resize_factor of PreProcBase (preprocUtils.cc:120-137) on the device against the restatement, bit for bit"""
    gray, bgr, smooth = raw_frames()
    for raw in (gray, bgr, smooth, gray.astype(np.float32) * 0.7 + 3.3):
        for kw in (dict(hist_eq=True), dict(resize_factor=0.5), dict(hist_eq=True, resize_factor=0.75), dict(hist_eq=True, ksize=0),
                   dict(resize_factor=1.6, ksize=0), dict(hist_eq=True, resize_factor=2.0)):
            gpu_ctx.preprocess(raw, **kw)
            want = P.preprocess(raw, **kw)
            got = gpu_ctx.get_image()
            assert got.shape == want.shape, (kw, got.shape, want.shape)
            assert np.array_equal(got, want), kw
    const = np.full((33, 47), 91, dtype=np.uint8)
    gpu_ctx.preprocess(const, hist_eq=True, ksize=0)
    assert np.all(gpu_ctx.get_image() == 91.0)
    with pytest.raises(mtf_amd.InvalidArgument):
        gpu_ctx.preprocess(gray, resize_factor=0.0)


@pytest.mark.gpu
def test_device_preprocess_matches_restatement(gpu_ctx):
    gray, bgr, smooth = raw_frames()
    for raw in (gray, bgr, smooth, gray.astype(np.float32), bgr.astype(np.float32)):
        gpu_ctx.preprocess(raw)
        got = gpu_ctx.get_image()
        want = P.preprocess(raw)
        assert got.shape == want.shape
        assert np.array_equal(got, want), np.abs(got - want).max()
        gpu_ctx.preprocess(raw, ksize=0)
        assert np.array_equal(gpu_ctx.get_image(), P.to_gray_f32(raw))
    # a strided view (a cropped ROI of a larger frame) goes through the row stride
    big = np.zeros((120, 150), dtype=np.uint8); big[10:107, 5:136] = gray
    gpu_ctx.preprocess(big[10:107, 5:136])
    assert np.array_equal(gpu_ctx.get_image(), P.preprocess(gray))
    import mtf_amd
    with pytest.raises(mtf_amd.FunctionNotImplemented):
        gpu_ctx.preprocess(gray, ksize=7)
    with pytest.raises(mtf_amd.InvalidArgument):
        gpu_ctx.preprocess(gray.astype(np.float64))


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [0.5, 0.7])
def test_device_pyramid_levels_and_pyramidal_tracker(gpu_ctx, scale):
    import mtf_amd
    from mtf_amd import _lib as L
    from mtf_amd.sm import LKTracker, PyramidalTracker
    f0 = synth.make_frame(480, 640)
    p_true = synth.random_small_homography(np.random.default_rng(8), 1.0)
    p_true[2] += 6.0; p_true[5] -= 4.0            # a displacement the single-level tracker struggles with
    f1 = synth.warp_frame(f0, p_true, (320.0, 240.0))
    gpu_ctx.set_image(f0)
    lvl = mtf_amd.Context(0)
    r, c = int(480 * scale), int(640 * scale)
    lvl.pyramid_level_from(gpu_ctx, r, c, pyr_down=(scale == 0.5))
    assert np.array_equal(lvl.get_image(), P.pyramid_level(f0, r, c, use_pyr_down=(scale == 0.5)))
    if scale == 0.5:
        with pytest.raises(mtf_amd.InvalidArgument):
            lvl.pyramid_level_from(gpu_ctx, 100, 100, pyr_down=True)     # cv::pyrDown's size assertion
    lvl.close()
    corners = synth.square_corners(320, 240, 120)
    pt = PyramidalTracker(gpu_ctx, lambda ctx, k: LKTracker(ctx, L.SM_ESM, L.SSM_HOMOGRAPHY, 40, 40, 1, host_solve=False,
                                                         max_iters=30, epsilon=1e-4), no_of_levels=3, scale_factor=scale)
    pt.initialize(corners)
    gpu_ctx.set_image(f1)
    out = pt.update()[0]
    W = synth.homography_from_state(p_true)
    q = W @ np.vstack([corners - np.array([[320.0], [240.0]]), np.ones(4)])
    want = q[:2] / q[2] + np.array([[320.0], [240.0]])
    assert np.abs(out - want).max() < 0.3
    assert pt.sizes[1] == (int(480 * scale), int(640 * scale))
