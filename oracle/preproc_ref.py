"""NumPy float32 restatement of the OpenCV operations MTF's pre-processor and PyramidalTracker call
(Utilities/src/preprocUtils.cc:108-127, Utilities/include/mtf/Utilities/preprocUtils.h:67-73, SM/src/PyramidalTracker.cc:88-97).

TEST INFRASTRUCTURE ONLY (tests/ import it; the product never does).

PARITY UNPINNED: OpenCV is a third-party dependency of the reference ("2.4.13, 3.3.0, 3.4.1 tested", ReadMe.md:112-116)
that is absent from /root/reference and from this image, so this file restates its published float32 algorithms:
  * cvtColor(BGR2GRAY) on CV_32F: gray = B*0.114f + G*0.587f + R*0.299f
  * getGaussianKernel(n, sigma > 0, CV_32F): exp(-x^2 / (2 sigma^2)) cast to float, normalised by the double sum of the floats
  * sepFilter2D with a symmetric 5-tap kernel: rows S[0]*k0 + (S[-1]+S[1])*k1 + (S[-2]+S[2])*k2 (SymmRowSmallFilter), columns
    s = k0*S0; s += k1*(S+1 + S-1); s += k2*(S+2 + S-2) (SymmColumnFilter); BORDER_REFLECT_101
  * pyrDown: [1 4 6 4 1] x [1 4 6 4 1] / 256 with the same pair-sum order, BORDER_REFLECT_101
  * resize(INTER_LINEAR): fx = (float)((dx + 0.5) * scale - 0.5), floor, clamp at both ends, float weights
  * convertTo(CV_8U): cvRound (to nearest even) + saturation; equalizeHist: 256-bin histogram -> look-up table as below
Every intermediate is rounded to float32 exactly where OpenCV's float code rounds."""
import numpy as np

f32 = np.float32


def to_gray_f32(raw):
    raw = np.asarray(raw)
    if raw.ndim == 2:
        return raw.astype(f32)
    b, g, r = (raw[..., k].astype(f32) for k in range(3))
    return ((b * f32(0.114) + g * f32(0.587)).astype(f32) + r * f32(0.299)).astype(f32)


def gaussian_kernel5(sigma):
    x = np.arange(5, dtype=np.float64) - 2.0
    cf = np.exp((-0.5 / (sigma * sigma)) * x * x).astype(f32)
    s = 1.0 / float(np.sum(cf.astype(np.float64)))
    return (cf.astype(np.float64) * s).astype(f32)


def _reflect101(idx, n):
    idx = np.asarray(idx).copy()
    if n == 1:
        return np.zeros_like(idx)
    while ((idx < 0) | (idx >= n)).any():
        idx = np.where(idx < 0, -idx, idx)
        idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
    return idx


def sym5(img, kx, ky):
    img = np.asarray(img, dtype=f32)
    rows, cols = img.shape
    xs = np.arange(cols)
    S = lambda d: img[:, _reflect101(xs + d, cols)]
    t = (img * kx[2] + (S(-1) + S(1)).astype(f32) * kx[3]).astype(f32)
    t = (t + (S(-2) + S(2)).astype(f32) * kx[4]).astype(f32)
    ys = np.arange(rows)
    R = lambda d: t[_reflect101(ys + d, rows), :]
    s = (ky[2] * t + f32(0)).astype(f32)
    s = (s + ky[3] * (R(1) + R(-1)).astype(f32)).astype(f32)
    s = (s + ky[4] * (R(2) + R(-2)).astype(f32)).astype(f32)
    return s


def gaussian_blur5(img, sigma_x=3.0, sigma_y=0.0):
    kx = gaussian_kernel5(sigma_x)
    ky = gaussian_kernel5(sigma_y if sigma_y > 0 else sigma_x)
    return sym5(img, kx, ky)


def to_u8(img):
    """Mat::convertTo(CV_8U) of a float image: cvRound (round half to even) + saturate_cast<uchar>"""
    return np.clip(np.rint(np.asarray(img, dtype=f32)), 0, 255).astype(np.uint8)


def equalize_hist_u8(src):
    """cv::equalizeHist (imgproc/histogram.cpp, OpenCV 2.4 / 3.x): hist of the 256 levels; i = first occupied level; an image
    with one level only is set to that level; else scale = 255.f / (total - hist[i]) (float), lut[i] = 0 and
    lut[j] = saturate_cast<uchar>(sum_{i < k <= j} hist[k] * scale) for j > i (int sum times float scale, cvRound)"""
    src = np.asarray(src, dtype=np.uint8)
    hist = np.bincount(src.ravel(), minlength=256)
    i = int(np.nonzero(hist)[0][0])
    total = src.size
    if hist[i] == total:
        return np.full_like(src, i)
    scale = f32(255.0) / f32(total - hist[i])
    lut = np.zeros(256, dtype=np.uint8)
    csum = np.cumsum(hist[i + 1:]).astype(np.int64)
    lut[i + 1:] = np.clip(np.rint((csum.astype(f32) * scale).astype(f32)), 0, 255).astype(np.uint8)
    return lut[src]


def preprocess(raw, ksize=5, sigma_x=3.0, sigma_y=0.0, hist_eq=False, resize_factor=1.0):
    """PreProcBase::processFrame, CV_32FC1 output (Utilities/src/preprocUtils.cc:108-137)"""
    g = to_gray_f32(raw)
    if hist_eq:      # :120-125  frame_gs.convertTo(uchar); equalizeHist; convertTo(float)
        g = equalize_hist_u8(to_u8(g)).astype(f32)
    if ksize != 0:   # apply(): GaussianSmoothing, preprocUtils.h:67-73
        g = gaussian_blur5(g, sigma_x, sigma_y)
    if resize_factor != 1:   # :127-129  cv::resize(frame_gs, frame_resized, frame_resized.size()), size from :63-64
        g = resize_linear(g, int(g.shape[0] * resize_factor), int(g.shape[1] * resize_factor))
    return g


def pyr_down(img, drows, dcols):
    img = np.asarray(img, dtype=f32)
    srows, scols = img.shape
    x2 = 2 * np.arange(dcols)
    C = lambda d: img[:, _reflect101(x2 + d, scols)]
    r = (C(0) * f32(6) + (C(-1) + C(1)).astype(f32) * f32(4)).astype(f32)
    r = ((r + C(-2)).astype(f32) + C(2)).astype(f32)
    y2 = 2 * np.arange(drows)
    R = lambda d: r[_reflect101(y2 + d, srows), :]
    o = (R(0) * f32(6) + (R(-1) + R(1)).astype(f32) * f32(4)).astype(f32)
    o = ((o + R(-2)).astype(f32) + R(2)).astype(f32)
    return (o * f32(1.0 / 256.0)).astype(f32)


def _lin_coords(dn, sn):
    scale = float(sn) / float(dn)
    f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(f32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(f32)).astype(f32)
    lo = s < 0
    f[lo] = 0; s[lo] = 0
    hi = s >= sn - 1
    f[hi] = 0; s[hi] = sn - 1
    return s, f


def resize_linear(img, drows, dcols):
    img = np.asarray(img, dtype=f32)
    srows, scols = img.shape
    sx, fx = _lin_coords(dcols, scols)
    sy, fy = _lin_coords(drows, srows)
    sx1 = np.minimum(sx + 1, scols - 1)
    last = sx >= scols - 1
    h = ((img[:, sx] * (f32(1) - fx)).astype(f32) + (img[:, sx1] * fx).astype(f32)).astype(f32)
    h[:, last] = img[:, sx[last]]
    sy1 = np.minimum(sy + 1, srows - 1)
    return ((h[sy, :] * (f32(1) - fy)[:, None]).astype(f32) + (h[sy1, :] * fy[:, None]).astype(f32)).astype(f32)


def pyramid_level(img, drows, dcols, use_pyr_down=True):
    if use_pyr_down:
        return pyr_down(img, drows, dcols)
    return gaussian_blur5(resize_linear(img, drows, dcols), 3.0)
