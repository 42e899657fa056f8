"""ctypes view of the C++ host layer -- the reference-language side of the drop-in boundary: libmtfhost.so = the PRODUCT (mtf::hip::HipAM /
HipSSM adapter classes, mtf::hip::LK / PF device drivers, mtf_amd/host/*.cpp), libmtfharness.so = the HARNESS (the reference's callers
mtf::nt::ESM / FCLK / ICLK / PF restated over the virtuals, the templated SearchMethod<AM, SSM> shape, and the C wrapper this module
binds, mtf_amd/host/harness/*.cpp)."""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmtfharness.so")   # the C wrapper lives with the restated callers; it links libmtfhost.so (the product)
PRODUCT_LIB_PATH = os.path.join(_HERE, "libmtfhost.so")
HOST_SRC = os.path.join(_HERE, "host")
_h = None


def build():
    subprocess.check_call(["make", "-C", HOST_SRC, "-s", "-B"])
    return LIB_PATH


def lib():
    global _h
    if _h is None:
        _lib.lib()   # loads libmtfhip.so (and torch's HIP runtime first, when present)
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmtfharness.so is missing: run __graft_entry__.build()")
        H = C.CDLL(LIB_PATH)
        H.mtfhost_last_error.restype = C.c_char_p
        H.mtfhost_create.restype = C.c_void_p
        H.mtfhost_create.argtypes = [C.c_int] * 6 + [C.c_double] + [C.c_int] * 4 + [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        for fn in ("mtfhost_destroy", "mtfhost_set_image", "mtfhost_initialize", "mtfhost_set_region", "mtfhost_update",
                   "mtfhost_get_region"):
            getattr(H, fn).argtypes = None
        H.mtfhost_set_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        H.mtfhost_initialize.argtypes = [C.c_void_p, C.c_void_p]
        H.mtfhost_set_region.argtypes = [C.c_void_p, C.c_void_p]
        H.mtfhost_update.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        H.mtfhost_get_region.argtypes = [C.c_void_p, C.c_void_p]
        H.mtfhost_destroy.argtypes = [C.c_void_p]
        H.mtfhost_qr_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        H.mtfhost_ssm_algebra.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        H.mtfhost_set_learning.argtypes = [C.c_void_p, C.c_int, C.c_double]
        H.mtfhost_dist_feat.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        H.mtfhost_pf_create.restype = C.c_void_p
        H.mtfhost_pf_create.argtypes = [C.c_int] * 7 + [C.c_double] + [C.c_int] * 6 + [C.c_void_p, C.c_double, C.c_ulonglong, C.c_int]
        H.mtfhost_pf_create_ex.restype = C.c_void_p
        H.mtfhost_pf_create_ex.argtypes = [C.c_int] * 7 + [C.c_double] + [C.c_int] * 6 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                           C.c_int, C.c_double, C.c_ulonglong, C.c_int]
        H.mtfhost_pf_create_pix.restype = C.c_void_p
        H.mtfhost_pf_create_pix.argtypes = H.mtfhost_pf_create_ex.argtypes + [C.c_void_p]
        H.mtfhost_templated_fclk.argtypes = [C.c_int] * 5 + [C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        H.mtfhost_ssm_random_walk.argtypes = [C.c_void_p, C.c_ulonglong, C.c_int, C.c_void_p, C.c_void_p]
        H.mtfhost_ssm_pts_after_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        H.mtfhost_grid_create.restype = C.c_void_p
        H.mtfhost_grid_create.argtypes = [C.c_int] * 12 + [C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        H.mtfhost_grid_destroy.argtypes = [C.c_void_p]
        H.mtfhost_grid_set_estimator.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        H.mtfhost_grid_call.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        H.mtfhost_grid_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _h = H
    return _h


class HostError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise HostError(lib().mtfhost_last_error().decode("utf-8", "replace"))


class CppTracker:
    """mtf::nt::{ESM,FCLK,ICLK} (or, with device_loop, mtf::hip::LK) over mtf::hip::{HipAM,HipSSM}; parameter names and defaults
    are the reference's (leven_marq defaults to true as in ESMParams.cc / FCLKParams.cc / ICLKParams.cc)."""

    def __init__(self, sm, am=_lib.AM_SSD, ssm=_lib.SSM_HOMOGRAPHY, resx=50, resy=50, max_iters=30, epsilon=1e-4,
                 jac_type=1, hess_type=-1, chained_warp=1, leven_marq=1, lm_delta_init=0.01, lm_delta_update=10.0,
                 device=0, sec_ord_hess=0, n_channels=1, device_loop=False):
        # device_loop: mtf::hip::LK instead of mtf::nt::ESM / FCLK / ICLK -- the same search method and parameters with the whole
        # update() loop behind one C-ABI call (mtfhip_batch_track) instead of a loop over the AM / SSM virtuals
        h = lib().mtfhost_create(sm + (16 if device_loop else 0), am, ssm, resx, resy, max_iters, epsilon, jac_type, hess_type, chained_warp,
                                 leven_marq, lm_delta_init, lm_delta_update, device, sec_ord_hess, n_channels)
        self.n_channels = n_channels
        if not h:
            raise HostError(lib().mtfhost_last_error().decode("utf-8", "replace"))
        self._h = C.c_void_p(h)
        self._img = None
        self.iters = 0

    def __del__(self):
        if getattr(self, "_h", None) and not sys.is_finalizing():
            lib().mtfhost_destroy(self._h)
            self._h = None

    def set_image(self, img):
        assert img.dtype == np.float32 and img.flags["C_CONTIGUOUS"]
        self._img = img   # borrowed, as TrackerBase::setImage does (include/mtf/TrackerBase.h:22-26)
        step = img.shape[1] * (img.shape[2] if img.ndim == 3 else 1)     # floats per row
        assert (img.shape[2] if img.ndim == 3 else 1) == getattr(self, "n_channels", 1)
        _check(lib().mtfhost_set_image(self._h, img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1], step))

    @staticmethod
    def _c(corners):
        return np.ascontiguousarray(np.asarray(corners, dtype=np.float64).reshape(2, 4).T.ravel())

    def initialize(self, corners):
        c = self._c(corners)
        _check(lib().mtfhost_initialize(self._h, c.ctypes.data_as(C.c_void_p)))

    def set_region(self, corners):
        c = self._c(corners)
        _check(lib().mtfhost_set_region(self._h, c.ctypes.data_as(C.c_void_p)))

    def update(self):
        n = C.c_int(0)
        _check(lib().mtfhost_update(self._h, C.byref(n)))
        self.iters = n.value
        return self.get_region()

    def dist_feat(self):
        """AppearanceModel::updateDistFeat + getDistFeat of the patch at the tracker's current state"""
        n = C.c_int(0)
        _check(lib().mtfhost_dist_feat(self._h, None, C.byref(n)))
        out = np.empty(n.value)
        _check(lib().mtfhost_dist_feat(self._h, out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out

    def set_learning(self, enable, learning_rate=0.5):
        """enable_learning of the search method + learning_rate of the AM (online template update after every update())"""
        _check(lib().mtfhost_set_learning(self._h, int(bool(enable)), C.c_double(learning_rate)))

    def ssm_algebra(self, what, a=None, b=None, n_out=None):
        """StateSpaceModel virtuals that are host algebra (see mtfhost_ssm_algebra)"""
        za = np.ascontiguousarray(a, dtype=np.float64) if a is not None else np.zeros(8)
        zb = np.ascontiguousarray(b, dtype=np.float64) if b is not None else np.zeros(8)
        out = np.zeros(n_out or 8)
        _check(lib().mtfhost_ssm_algebra(self._h, int(what), za.ctypes.data_as(C.c_void_p), zb.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p)))
        return out

    def pts_after_update(self, dp, eager, n_pts):
        """ssm->compositionalUpdate(dp), then the bytes behind ssm->getPts() read through the base class (2, n_pts)"""
        d = np.ascontiguousarray(np.asarray(dp, dtype=np.float64))
        out = np.empty((n_pts, 2))
        _check(lib().mtfhost_ssm_pts_after_update(self._h, d.ctypes.data_as(C.c_void_p), int(bool(eager)), out.ctypes.data_as(C.c_void_p)))
        return out.T.copy()

    def get_region(self):
        out = np.empty(8)
        _check(lib().mtfhost_get_region(self._h, out.ctypes.data_as(C.c_void_p)))
        return out.reshape(4, 2).T.copy()


class CppParticleFilter(CppTracker):
    """the particle filter of the C++ host layer: device_filter=True -> mtf::hip::PF (all particles of an iteration on the
    device, mtfhip_pf_*), False -> mtf::nt::PF, the literal loop of SM/src/NT/PF.cc over the AM / SSM virtuals"""

    def __init__(self, device_filter=True, am=_lib.AM_SSD, ssm=_lib.SSM_HOMOGRAPHY, resx=50, resy=50, n_particles=500, max_iters=1,
                 epsilon=0.01, dynamic_model=0, update_type=1, likelihood_func=0, resampling_type=1, mean_type=0,
                 corner_based_sampling=1, ssm_sigma=(0.01, 0.01, 2.0, 0.01, 0.01, 2.0, 1e-5, 1e-5), likelihood_alpha=1.0, seed=1,
                 device=0, ssm_mean=None, update_distr_wts=0, min_distr_wt=0.1, adaptive_resampling_thresh=0.0, jacobian_as_sigma=0, pix_sigma=None):
        """ssm_sigma: one row, or several rows = several sampler distributions (PFParams::processDistributions)"""
        rows = [list(ssm_sigma)] if np.ndim(ssm_sigma) == 1 else [list(r) for r in ssm_sigma]
        mrows = [[0.0] * 8 for _ in rows] if ssm_mean is None else ([list(ssm_mean)] if np.ndim(ssm_mean) == 1 else [list(r) for r in ssm_mean])
        while len(mrows) < len(rows): mrows.append(mrows[-1])
        sg = np.zeros((len(rows), 8)); mn = np.zeros((len(rows), 8))
        for i, r in enumerate(rows): sg[i, :min(8, len(r))] = r[:8]
        for i, r in enumerate(mrows[:len(rows)]): mn[i, :min(8, len(r))] = r[:8]
        px = None if pix_sigma is None else np.ascontiguousarray(np.atleast_1d(np.asarray(pix_sigma, dtype=np.float64)))
        if px is not None:
            sg = np.ones((len(px), 8)); mn = np.zeros((len(px), 8))
        h = lib().mtfhost_pf_create_pix(int(bool(device_filter)), am, ssm, resx, resy, n_particles, max_iters, epsilon, dynamic_model,
                                        update_type, likelihood_func, resampling_type, mean_type, int(bool(corner_based_sampling)),
                                        sg.shape[0], sg.ctypes.data_as(C.c_void_p), mn.ctypes.data_as(C.c_void_p), int(update_distr_wts),
                                        float(min_distr_wt), float(adaptive_resampling_thresh), int(jacobian_as_sigma), likelihood_alpha, seed, device,
                                        None if px is None else px.ctypes.data_as(C.c_void_p))
        if not h:
            raise HostError(lib().mtfhost_last_error().decode("utf-8", "replace"))
        self._h = C.c_void_p(h)
        self._img = None
        self.iters = 0
        self.n_channels = 1
        self.S = 8 if ssm == _lib.SSM_HOMOGRAPHY else 6

    def random_walk_samples(self, seed, n, sigma):
        """n draws of StateSpaceModel::compositionalRandomWalk from the current state (the SSM sampler virtuals)"""
        sg = np.ascontiguousarray(np.asarray(list(sigma) + [0.0] * 8, dtype=np.float64)[:8])
        out = np.empty((n, self.S))
        _check(lib().mtfhost_ssm_random_walk(self._h, seed, n, sg.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out


def qr_solve(A, b):
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    Af = np.ascontiguousarray(A.T.ravel())
    bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64))
    x = np.empty(n)
    _check(lib().mtfhost_qr_solve(n, Af.ctypes.data_as(C.c_void_p), bb.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p)))
    return x


def templated_fclk(frame0, frame1, corners, am=_lib.AM_SSD, ssm=_lib.SSM_HOMOGRAPHY, resx=40, resy=40, max_iters=10, epsilon=1e-4, hess_type=1, device=0):
    """FCLK<HipAM, HipSSM> in the reference's TEMPLATED search-method shape (harness/TemplatedSM.h: models by value, built from
    `const AM::ParamType *` / `const SSM::ParamType *`): initialize on frame0, one update() on frame1 -> (corners (2, 4), iterations)"""
    assert frame0.dtype == np.float32 and frame1.dtype == np.float32 and frame0.shape == frame1.shape and frame0.flags["C_CONTIGUOUS"] and frame1.flags["C_CONTIGUOUS"]
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.float64).reshape(2, 4).T.ravel())
    out = np.zeros(8)
    n = C.c_int(0)
    _check(lib().mtfhost_templated_fclk(am, ssm, resx, resy, max_iters, epsilon, hess_type, device, frame0.ctypes.data_as(C.c_void_p),
                                        frame1.ctypes.data_as(C.c_void_p), frame0.shape[0], frame0.shape[1], frame0.shape[1],
                                        c.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out.reshape(4, 2).T.copy(), n.value


def templated_sm(sm, frame0, frame1, corners, am=_lib.AM_SSD, ssm=_lib.SSM_HOMOGRAPHY, resx=40, resy=40, max_iters=10, epsilon=1e-4, jac_type=1,
                 hess_type=2, leven_marq=0, move=None, device=0):
    """ESM<HipAM, HipSSM> (sm = SM_ESM) / ICLK<HipAM, HipSSM> (sm = SM_ICLK) in the reference's templated shape (harness/TemplatedSM.h):
    initialize on frame0, update() on frame1; with move = (dx, dy) then setRegion(result + move) and a second update().
    -> (corners (2, 4), corners after the second update or None, [iterations, iterations])"""
    assert frame0.dtype == np.float32 and frame1.dtype == np.float32 and frame0.shape == frame1.shape and frame0.flags["C_CONTIGUOUS"] and frame1.flags["C_CONTIGUOUS"]
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.float64).reshape(2, 4).T.ravel())
    out, out2 = np.zeros(8), np.zeros(8)
    n = (C.c_int * 2)(0, 0)
    mv = np.ascontiguousarray(np.asarray(move, dtype=np.float64)) if move is not None else None
    fn = lib().mtfhost_templated_sm
    fn.argtypes = [C.c_int] * 6 + [C.c_double] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _check(fn(sm, am, ssm, resx, resy, max_iters, epsilon, jac_type, hess_type, leven_marq, device, frame0.ctypes.data_as(C.c_void_p),
              frame1.ctypes.data_as(C.c_void_p), frame0.shape[0], frame0.shape[1], frame0.shape[1], c.ctypes.data_as(C.c_void_p),
              mv.ctypes.data_as(C.c_void_p) if mv is not None else None, out.ctypes.data_as(C.c_void_p),
              out2.ctypes.data_as(C.c_void_p) if mv is not None else None, n))
    return out.reshape(4, 2).T.copy(), (out2.reshape(4, 2).T.copy() if mv is not None else None), [n[0], n[1]]


_GRID_EST = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_double))


class CppGridTracker:
    """mtf::hip::Grid (mtf_amd/host/DeviceGrid.h): GridTracker<SSM> with the reference's parameter block over one batch of patch
    trackers on the device.  estimator(prev_pts (n, 2), curr_pts (n, 2)) -> state update replaces the built-in least-squares fit."""

    def __init__(self, grid_size=10, patch_size=10, patch_sm=_lib.SM_ICLK, patch_am=_lib.AM_NCC, patch_ssm=_lib.SSM_AFFINE,
                 grid_ssm=_lib.SSM_HOMOGRAPHY, reset_at_each_frame=1, dyn_patch_size=0, patch_centroid_inside=1, max_iters=30, epsilon=1e-4,
                 hess_type=-1, leven_marq=0, device=0, estimator=None, grid_size_y=None, patch_size_y=None, fb_err_thresh=0.0, fb_reinit=1,
                 n_model_pts=4):
        gy, py = grid_size_y or grid_size, patch_size_y or patch_size
        h = lib().mtfhost_grid_create(grid_size, gy, patch_size, py, reset_at_each_frame, dyn_patch_size, patch_centroid_inside, patch_sm,
                                      patch_am, patch_ssm, grid_ssm, max_iters, epsilon, hess_type, leven_marq, device, float(fb_err_thresh),
                                      int(fb_reinit), int(n_model_pts))
        if not h:
            raise HostError(lib().mtfhost_last_error().decode("utf-8", "replace"))
        self._h = C.c_void_p(h)
        self.n, self.S = grid_size * gy, 8 if grid_ssm == _lib.SSM_HOMOGRAPHY else 6
        self._img, self._cb = None, None
        if estimator is not None:
            S = self.S

            def cb(_user, cnt, prev, curr, out):
                a = np.ctypeslib.as_array(prev, shape=(cnt, 2)).astype(np.float64)
                b = np.ctypeslib.as_array(curr, shape=(cnt, 2)).astype(np.float64)
                upd = np.asarray(estimator(a, b), dtype=np.float64)
                for i in range(S):
                    out[i] = upd[i]
            self._cb = _GRID_EST(cb)
            _check(lib().mtfhost_grid_set_estimator(self._h, C.cast(self._cb, C.c_void_p), None))

    def __del__(self):
        if getattr(self, "_h", None) and not sys.is_finalizing():
            lib().mtfhost_grid_destroy(self._h)
            self._h = None

    def set_image(self, img):
        assert img.dtype == np.float32 and img.flags["C_CONTIGUOUS"] and img.ndim == 2
        self._img = img
        _check(lib().mtfhost_grid_call(self._h, 0, None, img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1], img.shape[1]))

    def _corners_call(self, what, corners):
        c = np.ascontiguousarray(np.asarray(corners, dtype=np.float64).reshape(2, 4).T.ravel())
        _check(lib().mtfhost_grid_call(self._h, what, c.ctypes.data_as(C.c_void_p), None, 0, 0, 0))

    def initialize(self, corners):
        self._corners_call(1, corners)

    def set_region(self, corners):
        self._corners_call(3, corners)

    def update(self):
        _check(lib().mtfhost_grid_call(self._h, 2, None, None, 0, 0, 0))
        return self.get_region()

    def _get(self, what, size):
        out = np.empty(size)
        _check(lib().mtfhost_grid_get(self._h, what, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_region(self):
        return self._get(0, 8).reshape(4, 2).T.copy()

    def patch_corners(self):
        return self._get(1, 8 * self.n).reshape(self.n, 4, 2).transpose(0, 2, 1).copy()

    def patch_regions(self):
        return self._get(6, 8 * self.n).reshape(self.n, 4, 2).transpose(0, 2, 1).copy()

    def prev_pts(self):
        return self._get(2, 2 * self.n).reshape(self.n, 2)

    def curr_pts(self):
        return self._get(3, 2 * self.n).reshape(self.n, 2)

    def ssm_update(self):
        return self._get(4, self.S)

    def fb_prev_pts(self):
        return self._get(7, 2 * self.n).reshape(self.n, 2)

    def fb_err_mask(self):
        return self._get(8, self.n).astype(bool)

    def patch_iters(self):
        return self._get(5, self.n).astype(np.int32)

    def bench_video(self, frame_a, frame_b, n_frames=200):
        """the video loop on the C++ side: per frame setImage(the other of two frames) + update() -> (us per frame in update(), us per frame in
        setImage = the host-to-device copy of the frame)"""
        assert frame_a.dtype == np.float32 and frame_b.dtype == np.float32 and frame_a.shape == frame_b.shape and frame_a.flags["C_CONTIGUOUS"] and frame_b.flags["C_CONTIGUOUS"]
        us = (C.c_double * 2)()
        fn = lib().mtfhost_grid_bench_video
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _check(fn(self._h, int(n_frames), frame_a.ctypes.data_as(C.c_void_p), frame_b.ctypes.data_as(C.c_void_p), frame_a.shape[0], frame_a.shape[1],
                  frame_a.shape[1], C.cast(us, C.c_void_p)))
        return us[0], us[1]

    def bench_frames(self, region, n_frames=300, what=0):
        """microseconds per frame of a loop that runs on the C++ side: what = 0 mtfhip_grid_frame(region) calls (layout + setRegion + update,
        one launch), 1 mtf::hip::Grid::update() (the same + the estimator + the reset of the parameters)"""
        c = np.ascontiguousarray(np.asarray(region, dtype=np.float64).reshape(2, 4).T.ravel())
        us = C.c_double()
        fn = lib().mtfhost_grid_bench
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        _check(fn(self._h, int(what), int(n_frames), c.ctypes.data_as(C.c_void_p), C.byref(us)))
        return us.value
