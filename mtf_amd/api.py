"""Python mirror of the reference's AppearanceModel / StateSpaceModel interface over libmtfhip.so.

Method names and argument meaning follow the reference's virtuals (camelCase -> snake_case):
AM/include/mtf/AM/ImageBase.h:92-123, AM/include/mtf/AM/AppearanceModel.h:77-219,
SSM/include/mtf/SSM/StateSpaceModel.h:98-181.  NumPy arrays are in NumPy-natural orientation on
this side (pts (B, 2, N), grad (B, N, 2), J (B, N, S), H (B, S, S), corners (B, 2, 4)) and are
converted to/from the Eigen column-major layouts of the C ABI here.
"""
import sys
import atexit
import weakref
import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import (AM_MI, AM_NCC, AM_SSD, BUF_CURR_PTS, BUF_D2I0_DP2, BUF_D2I0_DX2, BUF_D2IM_DP2, BUF_D2IT_DP2, BUF_D2IT_DX2,
                   BUF_DF_DI0, BUF_DF_DIT, BUF_DI0_DX, BUF_DIT_DX, BUF_HESS_PTS,
                   BUF_GRAD_PTS, BUF_I0, BUF_INIT_PTS, BUF_IT, BUF_J0, BUF_JM, BUF_JT, JAC_APPROX, JAC_INIT,
                   JAC_PIX, JAC_WARPED, SM_ESM, SM_FCLK, SM_ICLK, SSM_AFFINE, SSM_HOMOGRAPHY, PatchDesc, SMDesc)


def _p(a):
    # (the address from the array interface: ~1.3 us against ~2.7 us for a.ctypes.data_as(...), which builds two helper objects per
    # call -- every wrapper below pays this once per array argument.  The caller keeps `a` alive across the C call: every use
    # passes a named local.)
    return C.c_void_p(a.__array_interface__["data"][0])


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def sm_desc(sm, **kw):
    """Class defaults of the reference (SM/src/ESMParams.cc:4-15, FCLKParams.cc:4-17, ICLKParams.cc:4-14)."""
    base = dict(sm=sm, jac_type=1, hess_type={SM_ESM: 2, SM_FCLK: 1, SM_ICLK: 0}[sm], chained_warp=1,
                materialize=1, max_iters=30, epsilon=1e-4, leven_marq=0, lm_delta_init=0.01, lm_delta_update=10.0,
                sec_ord_hess=0)
    base.update(kw)
    return SMDesc(**base)


_live_contexts = weakref.WeakSet()


def _close_all_contexts():
    # at interpreter exit, BEFORE the HIP runtime's own static destructors run: handles released in order (batches, then their
    # context) while the runtime is still whole
    for c in list(_live_contexts):
        try:
            c.close()
        except Exception:
            pass


atexit.register(_close_all_contexts)


class Context:
    """Device + stream + the current image (ImageBase::setCurrImg)."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        L.check(L.lib().mtfhip_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device
        self._img_keep = None
        self._batches = weakref.WeakSet()   # a batch holds a raw pointer to its context: close them first
        self._dependents = weakref.WeakSet()   # ... and whatever holds a raw pointer to a batch (the device particle filter) before those
        _live_contexts.add(self)

    def close(self):
        if self._h:
            for d in list(self._dependents):
                d.close()
            for b in list(self._batches):
                b.close()
            L.lib().mtfhip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if sys.is_finalizing():    # the HIP runtime may already be gone; the OS reclaims the device memory
            return
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        L.check(L.lib().mtfhip_ctx_synchronize(self._h))

    @property
    def stream(self):
        return L.lib().mtfhip_ctx_stream(self._h)

    def set_image(self, img):
        """Upload a host float32 H x W image (the CV_32FC1 input of AM/src/ImageBase.cc:38-60)."""
        img = np.asarray(img)
        if img.dtype == np.float32 and img.ndim == 3 and img.shape[2] == 3:     # CV_32FC3 for MCSSD / MCNCC / MCMI
            img = np.ascontiguousarray(img)
            L.check(L.lib().mtfhip_image_upload_mc(self._h, _p(img), img.shape[0], img.shape[1], img.shape[1] * 3, 3))
            return
        if img.dtype != np.float32 or img.ndim != 2:
            raise L.InvalidArgument(-1, "Input image type does not match the required type: 32FC1 / 32FC3")
        stride = img.strides[0] // 4
        if img.strides[1] != 4:
            img = np.ascontiguousarray(img)
            stride = img.shape[1]
        L.check(L.lib().mtfhip_image_upload(self._h, _p(img), img.shape[0], img.shape[1], stride))

    def preprocess(self, raw, ksize=5, sigma_x=3.0, sigma_y=0.0, hist_eq=False, resize_factor=1.0):
        """PreProcBase::update with GaussianSmoothing (preprocUtils.h:20-73, preprocUtils.cc:108-137): raw uint8 / float32 frame,
        H x W or H x W x 3 (BGR) -> gray float32 -> [hist_eq: 8 bit, cv::equalizeHist] -> Gaussian ksize x ksize ->
        [resize_factor: cv::resize] -> the current image, all on the device."""
        raw = np.asarray(raw)
        if raw.dtype not in (np.uint8, np.float32) or raw.ndim not in (2, 3) or (raw.ndim == 3 and raw.shape[2] != 3):
            raise L.InvalidArgument(-1, "PreProcBase::processFrame : Invalid input image type provided")
        if not raw.flags["C_CONTIGUOUS"]:
            raw = np.ascontiguousarray(raw)
        ch = 1 if raw.ndim == 2 else 3
        L.check(L.lib().mtfhip_image_preprocess_ex(self._h, _p(raw), raw.shape[0], raw.shape[1], raw.strides[0], ch,
                                                   0 if raw.dtype == np.uint8 else 1, int(ksize), float(sigma_x), float(sigma_y),
                                                   1 if hist_eq else 0, float(resize_factor)))

    def pyramid_level_from(self, src, rows, cols, pyr_down=True):
        """this context's image = one pyramid level below `src`'s (PyramidalTracker::updateImagePyramid)"""
        L.check(L.lib().mtfhip_image_pyramid_level(self._h, src._h, int(rows), int(cols), 1 if pyr_down else 0))

    def image_shape(self):
        r, c = C.c_int(), C.c_int()
        L.check(L.lib().mtfhip_image_shape(self._h, C.byref(r), C.byref(c)))
        return r.value, c.value

    def get_image(self):
        """PreProcBase::getFrame: read-back of the current float32 image"""
        r, c = self.image_shape()
        out = np.empty((r, c), dtype=np.float32)
        L.check(L.lib().mtfhip_image_download(self._h, _p(out), r, c))
        return out

    def keep_prev(self):
        """prev_img = curr_img.clone() (SM/src/GridTracker.cc:241-243, 266): the current image becomes the context's previous image
        (no copy for an uploaded / pre-processed image: the next frame goes to the other of two device buffers)"""
        L.check(L.lib().mtfhip_image_keep_prev(self._h))

    def has_prev(self):
        return bool(L.lib().mtfhip_image_has_prev(self._h))

    def swap_prev(self):
        """setImage(prev_img) / setImage(curr_img) (GridTracker.cc:300, 304): current and previous image change places"""
        L.check(L.lib().mtfhip_image_swap_prev(self._h))

    def set_image_device(self, dev_ptr, height, width, row_stride=None, keep=None):
        """Adopt a float32 image already resident in HBM (e.g. a torch tensor's data_ptr())."""
        self._img_keep = keep
        L.check(L.lib().mtfhip_image_borrow(self._h, C.c_void_p(dev_ptr), height, width, row_stride or width))

    def timing(self, on=True):
        """on: False/0 off, True/1 every launch, n > 1 every n-th launch of a kernel family"""
        L.check(L.lib().mtfhip_timing_enable(self._h, int(on)))

    def timing_reset(self):
        L.check(L.lib().mtfhip_timing_reset(self._h))

    def timing_get(self, family):
        avg, n = C.c_double(), C.c_int()
        L.check(L.lib().mtfhip_timing_get(self._h, family.encode(), C.byref(avg), C.byref(n)))
        return avg.value, n.value

    def timing_get_busy(self, family):
        """(ms during which at least one launch of the family was executing, launches): the union of their intervals"""
        busy, n = C.c_double(), C.c_int()
        L.check(L.lib().mtfhip_timing_get_busy(self._h, family.encode(), C.byref(busy), C.byref(n)))
        return busy.value, n.value


class Batch:
    """B independent targets (AM + SSM pairs) sharing the context's current image."""

    def __init__(self, ctx, am=AM_SSD, ssm=SSM_HOMOGRAPHY, resx=50, resy=50, n_targets=1, grad_eps=1e-8,
                 likelihood_alpha=1.0, mi_n_bins=8, mi_pre_seed=10.0, mi_pou=0, hess_eps=1.0, n_channels=1):
        self.ctx = ctx
        self.desc = PatchDesc(am, ssm, resx, resy, grad_eps, likelihood_alpha, mi_n_bins, mi_pre_seed, mi_pou, hess_eps,
                              n_channels)
        self._h = C.c_void_p()
        L.check(L.lib().mtfhip_batch_create(ctx._h, C.byref(self.desc), int(n_targets), C.byref(self._h)))
        ctx._batches.add(self)
        self.B = n_targets
        self.C = n_channels
        self.NP = resx * resy            # sample points (ImageBase::getNPix)
        self.N = self.NP * n_channels    # rows of every per-pixel AM array (getPatchSize); == NP for single channel
        self.S = 8 if ssm == SSM_HOMOGRAPHY else 6

    def close(self):
        if self._h:
            L.lib().mtfhip_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        if sys.is_finalizing():    # the HIP runtime may already be gone; the OS reclaims the device memory
            return
        try:
            self.close()
        except Exception:
            pass

    def set_math_mode(self, mode):
        """MATH_REPLAY (the reference's rounding, bit for bit) or MATH_FAST (FMA / reciprocal / closed-form gradient) for
        the kernels that do not materialise interface arrays; see include/mtfhip.h"""
        L.check(L.lib().mtfhip_batch_set_math_mode(self._h, int(mode)))

    def get_math_mode(self):
        return L.lib().mtfhip_batch_get_math_mode(self._h)

    # ---------------------------------------------------------- buffers
    _PER = {BUF_I0: 1, BUF_IT: 1, BUF_DF_DI0: 1, BUF_DF_DIT: 1}

    def read(self, buf):
        """Lazy read-back of a device buffer in NumPy-natural orientation."""
        B, N, S, NP = self.B, self.N, self.S, self.NP
        sizes = {BUF_I0: N, BUF_IT: N, BUF_DI0_DX: 2 * N, BUF_DIT_DX: 2 * N, BUF_DF_DI0: N, BUF_DF_DIT: N,
                 BUF_J0: N * S, BUF_JT: N * S, BUF_JM: N * S, BUF_INIT_PTS: 2 * NP, BUF_CURR_PTS: 2 * NP,
                 BUF_GRAD_PTS: 8 * NP, 12: NP, 13: NP, 14: 2 * NP, 15: 2 * NP, BUF_D2I0_DX2: 4 * N, BUF_D2IT_DX2: 4 * N,
                 BUF_HESS_PTS: 16 * NP, BUF_D2I0_DP2: S * S * N, BUF_D2IT_DP2: S * S * N, BUF_D2IM_DP2: S * S * N}
        out = np.empty((B, sizes[buf]))
        L.check(L.lib().mtfhip_batch_read(self._h, buf, _p(out)))
        if buf in (BUF_DI0_DX, BUF_DIT_DX):
            return out.reshape(B, 2, N).transpose(0, 2, 1)
        if buf in (BUF_J0, BUF_JT, BUF_JM):
            return out.reshape(B, S, N).transpose(0, 2, 1)
        if buf in (BUF_INIT_PTS, BUF_CURR_PTS, 14, 15):
            return out.reshape(B, NP, 2).transpose(0, 2, 1)
        if buf == BUF_GRAD_PTS:
            return out.reshape(B, NP, 8)
        if buf == BUF_HESS_PTS:
            return out.reshape(B, NP, 16)
        if buf in (BUF_D2I0_DX2, BUF_D2IT_DX2):
            return out.reshape(B, N, 2, 2)
        if buf in (BUF_D2I0_DP2, BUF_D2IT_DP2, BUF_D2IM_DP2):      # planes [c][r][N] -> (B, N, r, c)
            return out.reshape(B, S, S, N).transpose(0, 3, 2, 1)
        return out

    def write(self, buf, arr):
        """Overwrite a device buffer from a NumPy-natural array (ImageBase setters, ImageBase.h:93-100)."""
        B, N, S = self.B, self.N, self.S
        a = _f64(arr)
        if buf in (BUF_DI0_DX, BUF_DIT_DX):
            a = np.ascontiguousarray(a.reshape(B, N, 2).transpose(0, 2, 1))
        elif buf in (BUF_J0, BUF_JT, BUF_JM):
            a = np.ascontiguousarray(a.reshape(B, N, S).transpose(0, 2, 1))
        elif buf in (BUF_INIT_PTS, BUF_CURR_PTS, 14, 15):
            a = np.ascontiguousarray(a.reshape(B, 2, self.NP).transpose(0, 2, 1))
        L.check(L.lib().mtfhip_batch_write(self._h, buf, _p(a)))

    def device_ptr(self, buf):
        return L.lib().mtfhip_batch_device_ptr(self._h, buf)

    # ---------------------------------------------------------- StateSpaceModel
    def _corners_in(self, corners):
        c = _f64(corners).reshape(self.B, 2, 4)
        return np.ascontiguousarray(c.transpose(0, 2, 1)).reshape(self.B, 8)

    @staticmethod
    def _corners_out(flat):
        return flat.reshape(-1, 4, 2).transpose(0, 2, 1).copy()

    def set_corners(self, corners):
        c = self._corners_in(corners)
        L.check(L.lib().mtfhip_ssm_set_corners(self._h, _p(c)))

    initialize = set_corners  # StateSpaceModel::initialize is an alias of setCorners (StateSpaceModel.h:117-121)

    def set_state(self, states):
        s = _f64(states).reshape(self.B, self.S)
        L.check(L.lib().mtfhip_ssm_set_state(self._h, _p(s)))

    def estimate_state_sigma(self, pix_sigma):
        """StateSpaceModel::estimateStateSigma (ProjectiveBase.cc:201-213): (B, S) sampler sigmas for a pixel sigma"""
        out = np.empty((self.B, self.S))
        L.check(L.lib().mtfhip_ssm_estimate_state_sigma(self._h, C.c_double(float(pix_sigma)), _p(out)))
        return out

    def compositional_update(self, dps):
        s = _f64(dps).reshape(self.B, self.S)
        L.check(L.lib().mtfhip_ssm_compositional_update(self._h, _p(s)))

    def invert_state(self, states):
        s = _f64(states).reshape(self.B, self.S)
        out = np.empty_like(s)
        L.check(L.lib().mtfhip_ssm_invert_state(self._h, _p(s), _p(out)))
        return out

    def update_grad_pts(self, grad_eps=None):
        L.check(L.lib().mtfhip_ssm_update_grad_pts(self._h, self.desc.grad_eps if grad_eps is None else grad_eps))

    def cmpt_pix_jacobian(self, variant, grad_buf, dst_buf):
        L.check(L.lib().mtfhip_ssm_cmpt_pix_jacobian(self._h, variant, grad_buf, dst_buf))

    def cmpt_init_pix_jacobian(self, grad_buf=BUF_DI0_DX, dst_buf=BUF_J0):
        self.cmpt_pix_jacobian(JAC_INIT, grad_buf, dst_buf)

    def cmpt_warped_pix_jacobian(self, grad_buf=BUF_DIT_DX, dst_buf=BUF_JT):
        self.cmpt_pix_jacobian(JAC_WARPED, grad_buf, dst_buf)

    def get_corners(self):
        g = self.__dict__.get("_gc")
        if g is None:     # per-frame call: the buffer, its ctypes pointer (3-4 us to build) and the bound function are made once
            out = np.empty((self.B, 8))
            g = self._gc = (out, _p(out), L.lib().mtfhip_ssm_get_corners)
        out, po, fn = g
        L.check(fn(self._h, po))
        return self._corners_out(out)   # (a copy)

    def get_state(self):
        out = np.empty((self.B, self.S))
        L.check(L.lib().mtfhip_ssm_get_state(self._h, _p(out)))
        return out

    def get_warp(self):
        out = np.empty((self.B, 9))
        L.check(L.lib().mtfhip_ssm_get_warp(self._h, _p(out)))
        return out.reshape(self.B, 3, 3)

    def get_pts(self):
        return self.read(BUF_CURR_PTS)

    def additive_update(self, state_updates):
        """ProjectiveBase::additiveUpdate (SSM/src/ProjectiveBase.cc:51-55)"""
        s = _f64(state_updates).reshape(self.B, self.S)
        L.check(L.lib().mtfhip_ssm_additive_update(self._h, _p(s)))

    def apply_warp_to_corners(self, corners, states):
        c = self._corners_in(corners)
        s = _f64(states).reshape(self.B, self.S)
        out = np.empty((self.B, 8))
        L.check(L.lib().mtfhip_ssm_apply_warp_to_corners(self._h, _p(c), _p(s), _p(out)))
        return self._corners_out(out)

    # ---------------------------------------------------------- ImageBase
    def _pts_arg(self, pts, per):
        if pts is None:
            return None, None
        a = _f64(pts)
        if per == 2:  # (B, 2, NP) -> interleaved
            a = np.ascontiguousarray(a.reshape(self.B, 2, self.NP).transpose(0, 2, 1))
        else:
            a = np.ascontiguousarray(a.reshape(self.B, self.NP, 8))
        return a, _p(a)

    def initialize_pix_vals(self, pts=None):
        keep, p = self._pts_arg(pts, 2)
        L.check(L.lib().mtfhip_am_initialize_pix_vals(self._h, p))

    def update_pix_vals(self, pts=None):
        keep, p = self._pts_arg(pts, 2)
        L.check(L.lib().mtfhip_am_update_pix_vals(self._h, p))

    def update_model(self, pts=None, learning_rate=0.5):
        """SSD / NCC::updateModel (AM/src/SSD.cc:49-75): online template update at pts (None = the current points)"""
        keep, p = self._pts_arg(pts, 2)
        L.check(L.lib().mtfhip_am_update_model(self._h, p, C.c_double(learning_rate)))

    def initialize_pix_grad(self, pts=None, warped=False):
        keep, p = self._pts_arg(pts, 8 if warped else 2)
        fn = L.lib().mtfhip_am_initialize_pix_grad_warped if warped else L.lib().mtfhip_am_initialize_pix_grad
        L.check(fn(self._h, p))

    def update_pix_grad(self, pts=None, warped=False):
        keep, p = self._pts_arg(pts, 8 if warped else 2)
        fn = L.lib().mtfhip_am_update_pix_grad_warped if warped else L.lib().mtfhip_am_update_pix_grad
        L.check(fn(self._h, p))

    # ---------------------------------------------------------- AppearanceModel
    def initialize_similarity(self):
        L.check(L.lib().mtfhip_am_initialize_similarity(self._h))

    def initialize_grad(self):
        L.check(L.lib().mtfhip_am_initialize_grad(self._h))

    def initialize_hess(self):
        L.check(L.lib().mtfhip_am_initialize_hess(self._h))

    def update_similarity(self, prereq_only=True):
        L.check(L.lib().mtfhip_am_update_similarity(self._h, int(prereq_only)))

    def update_curr_grad(self):
        L.check(L.lib().mtfhip_am_update_curr_grad(self._h))

    def update_init_grad(self):
        L.check(L.lib().mtfhip_am_update_init_grad(self._h))

    def get_similarity(self):
        out = np.empty(self.B)
        L.check(L.lib().mtfhip_am_get_similarity(self._h, _p(out)))
        return out

    def get_likelihood(self):
        out = np.empty(self.B)
        L.check(L.lib().mtfhip_am_get_likelihood(self._h, _p(out)))
        return out

    def _g(self, fn, *bufs):
        out = np.empty((self.B, self.S))
        L.check(fn(self._h, *bufs, _p(out)))
        return out

    def _H(self, fn, *bufs):
        out = np.empty((self.B, self.S, self.S))
        L.check(fn(self._h, *bufs, _p(out)))
        return out.transpose(0, 2, 1).copy()  # column-major -> [r, c]

    def cmpt_init_jacobian(self, j0=BUF_J0):
        return self._g(L.lib().mtfhip_am_cmpt_init_jacobian, j0)

    def cmpt_curr_jacobian(self, jt=BUF_JT):
        return self._g(L.lib().mtfhip_am_cmpt_curr_jacobian, jt)

    def cmpt_difference_of_jacobians(self, j0=BUF_J0, jt=BUF_JT):
        return self._g(L.lib().mtfhip_am_cmpt_difference_of_jacobians, j0, jt)

    def cmpt_init_hessian(self, j0=BUF_J0):
        return self._H(L.lib().mtfhip_am_cmpt_init_hessian, j0)

    def cmpt_curr_hessian(self, jt=BUF_JT):
        return self._H(L.lib().mtfhip_am_cmpt_curr_hessian, jt)

    def cmpt_self_hessian(self, jt=BUF_JT):
        return self._H(L.lib().mtfhip_am_cmpt_self_hessian, jt)

    def cmpt_sum_of_hessians(self, j0=BUF_J0, jt=BUF_JT):
        return self._H(L.lib().mtfhip_am_cmpt_sum_of_hessians, j0, jt)

    def mean_jacobian(self):
        L.check(L.lib().mtfhip_sm_mean_jacobian(self._h))

    # ---------------------------------------------------------- second order (sec_ord_hess)
    def update_hess_pts(self, hess_eps=None):
        """StateSpaceModel::updateHessPts / initializeHessPts (Homography.cc:829-875, Affine.cc:315-350)"""
        L.check(L.lib().mtfhip_ssm_update_hess_pts(self._h, float(self.desc.hess_eps if hess_eps is None else hess_eps)))

    def _pix_hess(self, fn, fn_warped, pts, hess_pts, warped):
        keep, pp = self._pts_arg(pts, 2)
        if not warped:
            L.check(fn(self._h, pp))
            return
        hp = None if hess_pts is None else np.ascontiguousarray(_f64(hess_pts).reshape(self.B, self.NP * 16))
        L.check(fn_warped(self._h, pp, None if hp is None else _p(hp)))

    def initialize_pix_hess(self, pts=None, hess_pts=None, warped=False):
        """ImageBase::initializePixHess(pts) or, with warped=True, (pts, hess_pts); None = device-resident"""
        self._pix_hess(L.lib().mtfhip_am_initialize_pix_hess, L.lib().mtfhip_am_initialize_pix_hess_warped, pts, hess_pts, warped)

    def update_pix_hess(self, pts=None, hess_pts=None, warped=False):
        self._pix_hess(L.lib().mtfhip_am_update_pix_hess, L.lib().mtfhip_am_update_pix_hess_warped, pts, hess_pts, warped)

    def cmpt_pix_hessian(self, variant, hess_buf, grad_buf, dst_buf):
        """cmpt{Init,,Warped,Approx}PixHessian by variant = JAC_*"""
        L.check(L.lib().mtfhip_ssm_cmpt_pix_hessian(self._h, variant, hess_buf, grad_buf, dst_buf))

    def mean_pix_hessian(self):
        L.check(L.lib().mtfhip_sm_mean_pix_hessian(self._h))

    def cmpt_init_hessian2(self, j0=BUF_J0, d2=BUF_D2I0_DP2):
        return self._H(L.lib().mtfhip_am_cmpt_init_hessian2, j0, d2)

    def cmpt_curr_hessian2(self, jt=BUF_JT, d2=BUF_D2IT_DP2):
        return self._H(L.lib().mtfhip_am_cmpt_curr_hessian2, jt, d2)

    def cmpt_self_hessian2(self, jt=BUF_JT, d2=BUF_D2IT_DP2):
        return self._H(L.lib().mtfhip_am_cmpt_self_hessian2, jt, d2)

    def cmpt_sum_of_hessians2(self, j0=BUF_J0, jt=BUF_JT, d20=BUF_D2I0_DP2, d2t=BUF_D2IT_DP2):
        return self._H(L.lib().mtfhip_am_cmpt_sum_of_hessians2, j0, jt, d20, d2t)

    # ---------------------------------------------------------- fused path
    def init_template(self, sm):
        L.check(L.lib().mtfhip_batch_init_template(self._h, C.byref(sm)))

    def set_region(self, corners, sm):
        """the search method's setRegion (ESM / FCLK-InitialSelf refresh J0 and the constant Hessian on the new grid)"""
        c = self._corners_in(corners)
        L.check(L.lib().mtfhip_batch_set_region(self._h, _p(c), C.byref(sm)))

    @property
    def inline_warp(self):
        """True when single-target launches carry the warp in the kernel arguments (mtfhip_batch_inline_warp)"""
        return bool(L.lib().mtfhip_batch_inline_warp(self._h))

    def track_targets_per_launch(self, sm):
        return L.lib().mtfhip_batch_track_targets_per_launch(self._h, C.byref(sm))

    def track_queues(self, sm):
        """queues the device-side loop keeps busy with independent chunks of targets (mtfhip_batch_track_queues)"""
        return L.lib().mtfhip_batch_track_queues(self._h, C.byref(sm))

    def iterate(self, sm):
        f = np.empty(self.B)
        g = np.empty((self.B, self.S))
        H = np.empty((self.B, self.S, self.S))
        L.check(L.lib().mtfhip_batch_iterate(self._h, C.byref(sm), _p(f), _p(g), _p(H)))
        return f, g, H.transpose(0, 2, 1).copy()

    def track(self, sm):
        g = self.__dict__.get("_tk")
        if g is None:     # (as get_corners: one update() per frame)
            n, c = np.empty(self.B, dtype=np.int32), np.empty((self.B, 8))
            g = self._tk = (n, c, _p(n), _p(c), L.lib().mtfhip_batch_track)
        n, c, pn, pc, fn = g
        L.check(fn(self._h, C.byref(sm), pn, pc))
        return n.copy(), self._corners_out(c)

    def track_trace(self, max_passes):
        """debug trace of the device-side loop on (max_passes > 0) / off (0): mtfhip_batch_track_trace"""
        L.check(L.lib().mtfhip_batch_track_trace(self._h, int(max_passes)))
        self._trace_cap = int(max_passes)

    def read_track_trace(self, n_iters):
        """per target the list of per-pass records of the last track() call -- dicts with H (S x S), g, dp (the state update
        applied), corners (2 x 4) after it, f, undo, lm_delta, has_H -- the shape of the CPU trackers' trace()"""
        cap, S = self._trace_cap, self.S
        raw = np.empty((self.B, cap, 96))
        L.check(L.lib().mtfhip_batch_track_trace_read(self._h, _p(raw)))
        out = []
        for t in range(self.B):
            recs = []
            for k in range(min(int(n_iters[t]), cap)):
                r = raw[t, k]
                recs.append(dict(H=r[:64].reshape(8, 8)[:S, :S].copy(), g=r[64:64 + S].copy(), dp=r[72:72 + S].copy(),
                                 corners=r[80:88].reshape(4, 2).T.copy(), f=float(r[88]), undo=bool(r[90]), lm_delta=float(r[91]),
                                 has_H=bool(r[92])))
            out.append(recs)
        return out

    def track_region(self, corners, sm):
        """setRegion(corners) + update() of one frame in one C-ABI call (mtfhip_batch_track_region)"""
        tr = getattr(self, "_tr", None)
        if tr is None:     # per-frame call: the argument buffers and their ctypes pointers are built once
            r, n, c = np.empty((self.B, 4, 2)), np.empty(self.B, dtype=np.int32), np.empty((self.B, 4, 2))
            tr = self._tr = (r, n, c, _p(r), _p(n), _p(c), L.lib().mtfhip_batch_track_region)
        r, n, c, pr, pn, pc, fn = tr
        r[...] = np.asarray(corners, dtype=np.float64).reshape(self.B, 2, 4).transpose(0, 2, 1)
        L.check(fn(self._h, C.byref(sm), pr, pn, pc))
        return n.copy(), c.transpose(0, 2, 1).copy()

    def grid_update(self, regions, sm):
        """GridTracker::update's patch half in one C-ABI call (mtfhip_grid_update): regions (B, 2, 4) -> iteration counts, corners
        (B, 2, 4), centroids (B, 2).  The returned arrays are this object's reused buffers: valid until the next call."""
        gu = getattr(self, "_gu", None)
        if gu is None:     # per-frame call: buffers, their pointers and the bound function are built once
            n, c, m = np.empty(self.B, dtype=np.int32), np.empty((self.B, 2, 4)), np.empty((self.B, 2))
            gu = self._gu = (n, c, m, _p(n), _p(c), _p(m), L.lib().mtfhip_grid_update, None)
        n, c, m, pn, pc, pm, fn, _ = gu
        r = regions if (type(regions) is np.ndarray and regions.dtype == np.float64 and regions.flags.c_contiguous and regions.size == 8 * self.B) \
            else np.ascontiguousarray(np.asarray(regions, dtype=np.float64).reshape(self.B, 2, 4))
        L.check(fn(self._h, C.byref(sm), r.ctypes.data, pn, pc, pm))
        return n, c, m

    def grid_frame(self, gd, sm, region=None):
        """one frame of GridTracker's patch half (mtfhip_grid_frame): with `region` (2 x 4) the patches are laid over it and reset
        (setRegion) in the same call, then every patch tracker updates.  -> iteration counts, corners (B, 2, 4), centroids (B, 2)
        float32 as the reference's cv::Point2f.  Reused buffers: valid until the next call."""
        gf = getattr(self, "_gf", None)
        if gf is None:
            n, c, m, r = np.empty(self.B, dtype=np.int32), np.empty((self.B, 4, 2)), np.empty((self.B, 2), dtype=np.float32), np.empty(8)
            gf = self._gf = (n, c, m, r, _p(n), _p(c), _p(m), _p(r), L.lib().mtfhip_grid_frame)
        n, c, m, r, pn, pc, pm, pr, fn = gf
        if region is not None:
            r[...] = np.asarray(region, dtype=np.float64).reshape(2, 4).T.ravel()
        L.check(fn(self._h, C.byref(sm), C.byref(gd), pr if region is not None else None, pn, pc, pm))
        return n, c.transpose(0, 2, 1), m

    def grid_reset(self, gd, sm, region, reinit):
        """GridTracker::resetTrackers(reinit) (mtfhip_grid_reset) -> patch corners (B, 2, 4), prev_pts (B, 2) float32"""
        pcs, pp = np.empty((self.B, 4, 2)), np.empty((self.B, 2), dtype=np.float32)
        r = np.ascontiguousarray(np.asarray(region, dtype=np.float64).reshape(2, 4).T)
        L.check(L.lib().mtfhip_grid_reset(self._h, C.byref(sm), C.byref(gd), _p(r), int(bool(reinit)), _p(pcs), _p(pp)))
        return pcs.transpose(0, 2, 1).copy(), pp

    def grid_backward(self, gd, sm, fb):
        """the patch loop of GridTracker::backwardEstimation (mtfhip_grid_backward) -> iteration counts, the regions reached on the
        previous frame (B, 2, 4), fb_prev_pts (B, 2) float32"""
        n, c, m = np.empty(self.B, dtype=np.int32), np.empty((self.B, 4, 2)), np.empty((self.B, 2), dtype=np.float32)
        L.check(L.lib().mtfhip_grid_backward(self._h, C.addressof(sm), C.addressof(gd), C.addressof(fb), _p(n), _p(c), _p(m)))
        return n, c.transpose(0, 2, 1).copy(), m

    def grid_frame_fb(self, gd, sm, fb, prev_pts, region=None):
        """GridTracker::update's patch loop + backwardEstimation (mtfhip_grid_frame_fb) -> dict(n_iters, corners (B, 2, 4), centroids,
        fb_prev_pts, fb_err_mask (B,) bool, prev_masked (c, 2), curr_masked (c, 2))"""
        B = self.B
        n, c, m = np.empty(B, dtype=np.int32), np.empty((B, 4, 2)), np.empty((B, 2), dtype=np.float32)
        fbp, mask = np.empty((B, 2), dtype=np.float32), np.empty(B, dtype=np.uint8)
        pm, cm, cnt = np.empty((B, 2), dtype=np.float32), np.empty((B, 2), dtype=np.float32), C.c_int()
        pp = np.ascontiguousarray(prev_pts, dtype=np.float32).reshape(B, 2)
        r = None if region is None else np.ascontiguousarray(np.asarray(region, dtype=np.float64).reshape(2, 4).T)
        L.check(L.lib().mtfhip_grid_frame_fb(self._h, C.addressof(sm), C.addressof(gd), C.addressof(fb), None if r is None else _p(r), _p(pp), _p(n), _p(c), _p(m),
                                             _p(fbp), _p(mask), _p(pm), _p(cm), C.addressof(cnt)))
        k = cnt.value
        return dict(n_iters=n, corners=c.transpose(0, 2, 1).copy(), centroids=m, fb_prev_pts=fbp, fb_err_mask=mask.astype(bool), prev_masked=pm[:k].copy(),
                    curr_masked=cm[:k].copy())

    # ---------------------------------------------------------- NN dataset generation
    def nn_feature_size(self):
        """am->getDistFeatSize(): N (SSD, NCC), 5 N (MI)"""
        f = C.c_int()
        L.check(L.lib().mtfhip_nn_feature_size(self._h, C.addressof(f)))
        return f.value

    def nn_desc(self, n_samples, sigma, mean=None, seed=0):
        d = L.NnDesc()
        d.n_samples, d.additive_update, d.seed = int(n_samples), 0, int(seed)
        sg = np.zeros(8); sg[:self.S] = np.asarray(sigma, dtype=np.float64)[:self.S]
        mn = np.zeros(8)
        if mean is not None:
            mn[:self.S] = np.asarray(mean, dtype=np.float64)[:self.S]
        for k in range(8):
            d.sigma[k] = sg[k]; d.mean[k] = mn[k]
        return d

    def nn_dataset(self, n_samples, sigma, mean=None, seed=0, perturbations=None):
        """NN::generateDataset (mtfhip_nn_dataset): perturbations given (n_samples, S) or drawn on the device -> (perturbations, features
        (n_samples, feat_size)) on the host"""
        d = self.nn_desc(n_samples, sigma, mean, seed)
        F = self.nn_feature_size()
        feat, pout = np.empty((n_samples, F)), np.empty((n_samples, self.S))
        pin = None if perturbations is None else np.ascontiguousarray(np.asarray(perturbations, dtype=np.float64).reshape(n_samples, self.S))
        L.check(L.lib().mtfhip_nn_dataset(self._h, C.addressof(d), None if pin is None else _p(pin), _p(pout), _p(feat)))
        return pout, feat

    def nn_dataset_dev(self, desc, dev_features_ptr, row_lo, row_count, dev_perts_in_ptr=None, dev_perts_out_ptr=None):
        """rows [row_lo, row_lo + row_count) of the dataset into device memory (mtfhip_nn_dataset_dev); pointers are device addresses"""
        L.check(L.lib().mtfhip_nn_dataset_dev(self._h, C.addressof(desc), C.c_void_p(dev_perts_in_ptr) if dev_perts_in_ptr else None,
                                              C.c_void_p(dev_perts_out_ptr) if dev_perts_out_ptr else None, C.c_void_p(dev_features_ptr), int(row_lo), int(row_count)))

    # ---------------------------------------------------------- candidate scoring
    def score_candidates(self, states, want_similarity=False):
        s = _f64(states).reshape(-1, self.S)
        lik = np.empty(s.shape[0])
        sim = np.empty(s.shape[0]) if want_similarity else None
        L.check(L.lib().mtfhip_score_candidates(self._h, _p(s), s.shape[0], _p(lik), _p(sim) if want_similarity else None))
        return (lik, sim) if want_similarity else lik

    def sample_candidates(self, states):
        """NN dataset rows: (C, N) distance features of the patches sampled under the C states."""
        s = _f64(states).reshape(-1, self.S)
        out = np.empty((s.shape[0], self.N))
        L.check(L.lib().mtfhip_sample_candidates(self._h, _p(s), s.shape[0], _p(out)))
        return out

    def score_candidates_dev(self, dev_states, n, dev_lik, dev_sim=None):
        L.check(L.lib().mtfhip_score_candidates_dev(self._h, C.c_void_p(dev_states), int(n), C.c_void_p(dev_lik),
                                                    C.c_void_p(dev_sim) if dev_sim else None))


# ---- SSM functions that are 3 x 3 algebra on the host (no device, no context): ProjectiveBase.cc:142-160,321-331,
# Homography.cc:877-883, Affine.cc:352-357,382-393 ----
def _state_size(ssm):
    return 8 if ssm == L.SSM_HOMOGRAPHY else 6


def identity_warp(ssm):
    out = np.empty(_state_size(ssm))
    L.check(L.lib().mtfhip_ssm_identity_warp(ssm, _p(out)))
    return out


def compose_warps(ssm, state_1, state_2):
    """state of W(state_2) * W(state_1), read back as the reference does (no renormalisation)"""
    a, b, out = _f64(state_1).reshape(-1), _f64(state_2).reshape(-1), np.empty(_state_size(ssm))
    L.check(L.lib().mtfhip_ssm_compose_warps(ssm, _p(a), _p(b), _p(out)))
    return out


def estimate_warp_from_corners(ssm, in_corners, out_corners):
    """in_corners, out_corners: (2, 4).  Homography: the 4-point DLT; Affine: the least-squares affine map."""
    a = np.ascontiguousarray(_f64(in_corners).reshape(2, 4).T); b = np.ascontiguousarray(_f64(out_corners).reshape(2, 4).T)
    out = np.empty(_state_size(ssm))
    L.check(L.lib().mtfhip_ssm_estimate_warp_from_corners(ssm, _p(a), _p(b), _p(out)))
    return out


def apply_warp_to_pts(ssm, pts, state):
    """pts: (2, n) -> (2, n)"""
    a = np.ascontiguousarray(_f64(pts).reshape(2, -1).T)
    st = _f64(state).reshape(-1)
    out = np.empty_like(a)
    L.check(L.lib().mtfhip_ssm_apply_warp_to_pts(ssm, _p(a), a.shape[0], _p(st), _p(out)))
    return out.T.copy()

