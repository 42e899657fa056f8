"""Deterministic synthetic frames and target regions.

The reference ships no images; its own synthetic-sequence tool warps a real frame
with known SSM perturbations (Examples/cpp/generateSyntheticSeq.cc:1-140).  Here the
base frame is a band-limited texture (sum of random 2-D sinusoids) passed through a
5x5 sigma=3 Gaussian, mimicking the reference's default pre-processor
(Config/include/mtf/Config/parameters.h:229-235), stored as contiguous float32
(the CV_32FC1 input ImageBase expects, AM/src/ImageBase.cc:55-59).
"""
import numpy as np

DEFAULT_SEED = 20260928


def _gauss_kernel(ksize=5, sigma=3.0):
    r = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(r * r) / (2.0 * sigma * sigma))
    return k / k.sum()


def _blur(img, ksize=5, sigma=3.0):
    k = _gauss_kernel(ksize, sigma)
    pad = ksize // 2
    p = np.pad(img, pad, mode="reflect")
    tmp = np.zeros_like(img)
    for i in range(ksize):
        tmp += k[i] * p[pad:-pad, i:i + img.shape[1]]
    p = np.pad(tmp, ((pad, pad), (0, 0)), mode="reflect")
    out = np.zeros_like(img)
    for i in range(ksize):
        out += k[i] * p[i:i + img.shape[0], :]
    return out


def make_frame(h=1024, w=1024, seed=DEFAULT_SEED, n_waves=32):
    """float32 H x W frame with values in [16, 240]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    img = np.zeros((h, w), dtype=np.float64)
    for _ in range(n_waves):
        period = rng.uniform(8.0, 128.0)
        theta = rng.uniform(0.0, 2.0 * np.pi)
        phase = rng.uniform(0.0, 2.0 * np.pi)
        amp = rng.uniform(0.3, 1.0)
        kx = 2.0 * np.pi * np.cos(theta) / period
        ky = 2.0 * np.pi * np.sin(theta) / period
        img += amp * np.sin(kx * xx + ky * yy + phase)
    img -= img.min()
    img *= (240.0 - 16.0) / img.max()
    img += 16.0
    img = _blur(img)
    return np.ascontiguousarray(img.astype(np.float32))


def make_frame_mc(h=512, w=512, seed=DEFAULT_SEED, channels=3):
    """float32 H x W x C frame (CV_32FC3 layout, interleaved): one independent texture per channel."""
    return np.ascontiguousarray(np.stack([make_frame(h, w, seed + 101 * c) for c in range(channels)], axis=2))


def homography_from_state(p):
    """3x3 warp of the reference's 8-dof parameterisation (SSM/src/Homography.cc:94-107)."""
    p = np.asarray(p, dtype=np.float64)
    return np.array([[1 + p[0], p[1], p[2]], [p[3], 1 + p[4], p[5]], [p[6], p[7], 1.0]])


def bilinear_f64(img, x, y, overflow=128.0):
    """Vectorised float64 bilinear sampler with a constant border (used only to build frame t+1)."""
    h, w = img.shape
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    inside = (x >= 0) & (x < w - 1) & (y >= 0) & (y < h - 1)
    xs = np.where(inside, x, 0.0)
    ys = np.where(inside, y, 0.0)
    lx = xs.astype(np.int64)
    ly = ys.astype(np.int64)
    dx = xs - lx
    dy = ys - ly
    im = img.astype(np.float64)
    v = (im[ly, lx] * (1 - dx) * (1 - dy) + im[ly, lx + 1] * dx * (1 - dy) +
         im[ly + 1, lx] * (1 - dx) * dy + im[ly + 1, lx + 1] * dx * dy)
    return np.where(inside, v, overflow)


def warp_frame(img, p_true, centre):
    """Frame t+1: frame t seen through the homography W(p_true) acting about `centre`.

    A point q of the new frame shows frame t at W^{-1}(q - c) + c ... i.e. an object at x in
    frame t appears at W(x - c) + c in the new frame.
    """
    if img.ndim == 3:
        return np.ascontiguousarray(np.stack([warp_frame(np.ascontiguousarray(img[..., c]), p_true, centre)
                                              for c in range(img.shape[2])], axis=2))
    h, w = img.shape
    W = homography_from_state(p_true)
    Wi = np.linalg.inv(W)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    qx = xx - centre[0]
    qy = yy - centre[1]
    d = Wi[2, 0] * qx + Wi[2, 1] * qy + Wi[2, 2]
    sx = (Wi[0, 0] * qx + Wi[0, 1] * qy + Wi[0, 2]) / d + centre[0]
    sy = (Wi[1, 0] * qx + Wi[1, 1] * qy + Wi[1, 2]) / d + centre[1]
    return np.ascontiguousarray(bilinear_f64(img, sx, sy).astype(np.float32))


def square_corners(cx, cy, size, jitter=0.37):
    """2x4 corners (TL, TR, BR, BL) of an axis-aligned square, offset by a sub-pixel jitter
    so that no sample falls exactly on an integer coordinate."""
    half = size / 2.0
    x0, x1 = cx - half + jitter, cx + half + jitter
    y0, y1 = cy - half + jitter, cy + half + jitter
    return np.array([[x0, x1, x1, x0], [y0, y0, y1, y1]], dtype=np.float64)


def random_small_homography(rng, scale=1.0):
    s = np.array([0.02, 0.02, 2.0, 0.02, 0.02, 2.0, 1e-4, 1e-4]) * scale
    return rng.uniform(-1.0, 1.0, size=8) * s


def pf_candidate_states(rng, n, sigma=(0.01, 0.01, 2.0, 0.01, 0.01, 2.0, 1e-5, 1e-5)):
    """Seeded N(0, sigma) homography states for the PF scoring config."""
    return rng.normal(0.0, 1.0, size=(n, 8)) * np.asarray(sigma, dtype=np.float64)
