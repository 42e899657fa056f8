"""ctypes loader of libmtfhip.so (the C-ABI declared in include/mtfhip.h).

There is no fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTFHIP_LIB", os.path.join(_HERE, "libmtfhip.so"))
CSRC = os.path.join(_HERE, "csrc")

AM_SSD, AM_NCC, AM_MI = 0, 1, 2
SSM_HOMOGRAPHY, SSM_AFFINE = 0, 1
SM_ESM, SM_FCLK, SM_ICLK = 0, 1, 2
JAC_INIT, JAC_PIX, JAC_WARPED, JAC_APPROX = 0, 1, 2, 3
MATH_REPLAY, MATH_FAST = 0, 1
(BUF_I0, BUF_IT, BUF_DI0_DX, BUF_DIT_DX, BUF_DF_DI0, BUF_DF_DIT, BUF_J0, BUF_JT, BUF_JM,
 BUF_INIT_PTS, BUF_CURR_PTS, BUF_GRAD_PTS, BUF_INIT_Z, BUF_CURR_Z, BUF_INIT_HXY, BUF_CURR_HXY,
 BUF_D2I0_DX2, BUF_D2IT_DX2, BUF_HESS_PTS, BUF_D2I0_DP2, BUF_D2IT_DP2, BUF_D2IM_DP2) = range(22)


class MtfHipError(RuntimeError):
    """Mirror of mtf::utils::Exception (Utilities/include/mtf/Utilities/excpUtils.h:8-55)."""

    def __init__(self, code, msg):
        super().__init__("mtfhip error %d: %s" % (code, msg))
        self.code = code


class InvalidArgument(MtfHipError):
    pass


class FunctionNotImplemented(MtfHipError):
    pass


class LogicError(MtfHipError):
    pass


_ERR = {-1: InvalidArgument, -2: FunctionNotImplemented, -3: LogicError}


class PatchDesc(C.Structure):
    _fields_ = [("am", C.c_int), ("ssm", C.c_int), ("resx", C.c_int), ("resy", C.c_int),
                ("grad_eps", C.c_double), ("likelihood_alpha", C.c_double), ("mi_n_bins", C.c_int),
                ("mi_pre_seed", C.c_double), ("mi_partition_of_unity", C.c_int), ("hess_eps", C.c_double),
                ("n_channels", C.c_int)]


class SMDesc(C.Structure):
    _fields_ = [("sm", C.c_int), ("jac_type", C.c_int), ("hess_type", C.c_int), ("chained_warp", C.c_int),
                ("materialize", C.c_int), ("max_iters", C.c_int), ("epsilon", C.c_double),
                ("leven_marq", C.c_int), ("lm_delta_init", C.c_double), ("lm_delta_update", C.c_double),
                ("sec_ord_hess", C.c_int)]


class PFDesc(C.Structure):
    _fields_ = [("n_particles", C.c_int), ("max_iters", C.c_int), ("epsilon", C.c_double), ("dynamic_model", C.c_int),
                ("update_type", C.c_int), ("likelihood_func", C.c_int), ("resampling_type", C.c_int), ("mean_type", C.c_int),
                ("corner_based_sampling", C.c_int), ("reset_to_mean", C.c_int), ("measurement_sigma", C.c_double),
                ("ar_coeff", C.c_double), ("ssm_sigma", C.c_double * 8), ("ssm_mean", C.c_double * 8), ("seed", C.c_ulonglong),
                ("pt_based_sampling", C.c_int), ("adaptive_resampling_thresh", C.c_double), ("update_distr_wts", C.c_int),
                ("min_distr_wt", C.c_double)]


class GridDesc(C.Structure):
    """mtfhip_grid_desc = GridTrackerParams (SM/src/GridTracker.cc:20-94)"""
    _fields_ = [("grid_size_x", C.c_int), ("grid_size_y", C.c_int), ("patch_size_x", C.c_int), ("patch_size_y", C.c_int),
                ("reset_at_each_frame", C.c_int), ("dyn_patch_size", C.c_int), ("patch_centroid_inside", C.c_int)]


class GridFbDesc(C.Structure):
    """mtfhip_grid_fb_desc: GridTrackerParams::fb_err_thresh / fb_reinit (SM/src/GridTracker.cc:186-190, 294-343) and est_params.n_model_pts"""
    _fields_ = [("fb_err_thresh", C.c_double), ("fb_reinit", C.c_int), ("n_model_pts", C.c_int)]


class NnDesc(C.Structure):
    """mtfhip_nn_desc: NN::generateDataset's parameters (SM/src/NT/NN.cc:56-84, 131-191)"""
    _fields_ = [("n_samples", C.c_int), ("additive_update", C.c_int), ("sigma", C.c_double * 8), ("mean", C.c_double * 8), ("seed", C.c_ulonglong)]


# every exported symbol of include/mtfhip.h (tests check that the library exports all of them)
SYMBOLS = [
    "mtfhip_last_error", "mtfhip_device_count", "mtfhip_ctx_create", "mtfhip_ctx_destroy",
    "mtfhip_ctx_synchronize", "mtfhip_ctx_stream", "mtfhip_image_upload", "mtfhip_image_upload_mc", "mtfhip_image_borrow",
    "mtfhip_image_preprocess", "mtfhip_image_pyramid_level", "mtfhip_image_download", "mtfhip_image_shape",
    "mtfhip_batch_create", "mtfhip_batch_destroy", "mtfhip_batch_n_targets", "mtfhip_batch_n_pix", "mtfhip_batch_patch_size",
    "mtfhip_batch_state_size", "mtfhip_batch_set_math_mode", "mtfhip_batch_get_math_mode", "mtfhip_batch_read", "mtfhip_batch_write", "mtfhip_batch_device_ptr",
    "mtfhip_ssm_set_corners", "mtfhip_ssm_set_state", "mtfhip_ssm_compositional_update",
    "mtfhip_ssm_invert_state", "mtfhip_ssm_update_grad_pts", "mtfhip_ssm_cmpt_pix_jacobian",
    "mtfhip_ssm_get_corners", "mtfhip_ssm_get_init_corners", "mtfhip_ssm_get_state", "mtfhip_ssm_get_warp",
    "mtfhip_ssm_apply_warp_to_corners", "mtfhip_ssm_identity_warp", "mtfhip_ssm_compose_warps",
    "mtfhip_ssm_estimate_warp_from_corners", "mtfhip_ssm_apply_warp_to_pts", "mtfhip_ssm_additive_update",
    "mtfhip_am_initialize_pix_vals", "mtfhip_am_update_pix_vals", "mtfhip_am_update_model", "mtfhip_am_initialize_pix_grad",
    "mtfhip_am_update_pix_grad", "mtfhip_am_initialize_pix_grad_warped", "mtfhip_am_update_pix_grad_warped",
    "mtfhip_am_initialize_similarity", "mtfhip_am_initialize_grad", "mtfhip_am_initialize_hess",
    "mtfhip_am_update_similarity", "mtfhip_am_update_curr_grad", "mtfhip_am_update_init_grad",
    "mtfhip_am_get_similarity", "mtfhip_am_get_likelihood",
    "mtfhip_am_cmpt_init_jacobian", "mtfhip_am_cmpt_curr_jacobian", "mtfhip_am_cmpt_difference_of_jacobians",
    "mtfhip_am_cmpt_init_hessian", "mtfhip_am_cmpt_curr_hessian", "mtfhip_am_cmpt_self_hessian",
    "mtfhip_am_cmpt_sum_of_hessians", "mtfhip_sm_mean_jacobian",
    "mtfhip_ssm_update_hess_pts", "mtfhip_am_initialize_pix_hess", "mtfhip_am_update_pix_hess",
    "mtfhip_am_initialize_pix_hess_warped", "mtfhip_am_update_pix_hess_warped", "mtfhip_ssm_cmpt_pix_hessian",
    "mtfhip_sm_mean_pix_hessian", "mtfhip_am_cmpt_init_hessian2", "mtfhip_am_cmpt_curr_hessian2",
    "mtfhip_am_cmpt_self_hessian2", "mtfhip_am_cmpt_sum_of_hessians2",
    "mtfhip_batch_init_template", "mtfhip_batch_set_region", "mtfhip_batch_iterate", "mtfhip_batch_track", "mtfhip_batch_track_region", "mtfhip_grid_update",
    "mtfhip_grid_res", "mtfhip_grid_layout", "mtfhip_grid_frame", "mtfhip_grid_reset",
    "mtfhip_image_keep_prev", "mtfhip_image_has_prev", "mtfhip_image_swap_prev", "mtfhip_grid_backward", "mtfhip_grid_fb_mask", "mtfhip_grid_frame_fb",
    "mtfhip_batch_track_targets_per_launch",
    "mtfhip_score_candidates", "mtfhip_score_candidates_dev",
    "mtfhip_sample_candidates", "mtfhip_sample_candidates_dev", "mtfhip_nn_feature_size", "mtfhip_nn_dataset", "mtfhip_nn_dataset_dev",
    "mtfhip_pf_create", "mtfhip_pf_destroy", "mtfhip_pf_initialize", "mtfhip_pf_set_region", "mtfhip_pf_set_sampler",
    "mtfhip_pf_iteration", "mtfhip_pf_update", "mtfhip_pf_get_particles", "mtfhip_pf_set_particles", "mtfhip_pf_max_similarity",
    "mtfhip_pf_set_max_similarity", "mtfhip_pf_set_distributions", "mtfhip_pf_set_distr_draws", "mtfhip_pf_get_distributions", "mtfhip_comm_create_loopback", "mtfhip_pf_shard_bounds",
    "mtfhip_batch_track_trace", "mtfhip_batch_track_trace_read", "mtfhip_image_preprocess_ex",
    "mtfhip_comm_unique_id", "mtfhip_comm_create", "mtfhip_comm_destroy", "mtfhip_comm_rank", "mtfhip_comm_world",
    "mtfhip_allgather_scores", "mtfhip_pf_set_comm", "mtfhip_pf_set_exchange", "mtfhip_pf_exchange_export", "mtfhip_pf_exchange_connect", "mtfhip_comm_create_detached",
    "mtfhip_timing_enable", "mtfhip_timing_reset", "mtfhip_timing_get", "mtfhip_timing_get_busy", "mtfhip_ssm_estimate_state_sigma", "mtfhip_batch_track_queues", "mtfhip_batch_inline_warp",
]


def build(force=False):
    """Compile libmtfhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(_HERE, "..", "include", "mtfhip.h")]
    newest = max(os.path.getmtime(p) for p in srcs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-B", "-j%d" % min(8, os.cpu_count() or 1)])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libmtfhip.so is missing (%s); build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` -- there is no CPU fallback" % LIB_PATH)
        try:
            # PyTorch-ROCm bundles its own libamdhip64 (SONAME libamdhip64.so.7, requested by torch as the
            # unversioned name): loaded first, it also satisfies libmtfhip's libamdhip64.so.7 dependency and the
            # process ends up with ONE HIP runtime.  The other order loads two runtimes that fight over the device.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.mtfhip_last_error.restype = C.c_char_p
        L.mtfhip_ctx_stream.restype = C.c_void_p
        L.mtfhip_batch_device_ptr.restype = C.c_void_p
        L.mtfhip_pf_max_similarity.restype = C.c_double
        L.mtfhip_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.mtfhip_batch_create.argtypes = [C.c_void_p, C.POINTER(PatchDesc), C.c_int, C.POINTER(C.c_void_p)]
        L.mtfhip_image_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mtfhip_image_borrow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mtfhip_image_upload_mc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mtfhip_image_preprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_double]
        L.mtfhip_image_preprocess_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_double, C.c_double, C.c_int, C.c_double]
        L.mtfhip_image_pyramid_level.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.mtfhip_image_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.mtfhip_image_shape.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mtfhip_grid_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mtfhip_grid_res.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mtfhip_grid_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mtfhip_grid_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mtfhip_grid_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mtfhip_image_keep_prev.argtypes = [C.c_void_p]
        L.mtfhip_image_has_prev.argtypes = [C.c_void_p]
        L.mtfhip_image_swap_prev.argtypes = [C.c_void_p]
        L.mtfhip_grid_backward.argtypes = [C.c_void_p] * 7
        L.mtfhip_grid_fb_mask.argtypes = [C.c_int] + [C.c_void_p] * 8
        L.mtfhip_grid_frame_fb.argtypes = [C.c_void_p] * 14
        L.mtfhip_ssm_update_grad_pts.argtypes = [C.c_void_p, C.c_double]
        L.mtfhip_ssm_update_hess_pts.argtypes = [C.c_void_p, C.c_double]
        for fn in ("mtfhip_am_initialize_pix_hess", "mtfhip_am_update_pix_hess"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p]
        for fn in ("mtfhip_am_initialize_pix_hess_warped", "mtfhip_am_update_pix_hess_warped"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mtfhip_ssm_cmpt_pix_hessian.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        for fn in ("mtfhip_am_cmpt_init_hessian2", "mtfhip_am_cmpt_curr_hessian2", "mtfhip_am_cmpt_self_hessian2"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mtfhip_am_cmpt_sum_of_hessians2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.mtfhip_score_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mtfhip_score_candidates_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mtfhip_sample_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mtfhip_sample_candidates_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.mtfhip_nn_feature_size.argtypes = [C.c_void_p, C.c_void_p]
        L.mtfhip_nn_dataset.argtypes = [C.c_void_p] * 5
        L.mtfhip_nn_dataset_dev.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int]
        L.mtfhip_batch_device_ptr.argtypes = [C.c_void_p, C.c_int]
        L.mtfhip_batch_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mtfhip_batch_write.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.mtfhip_timing_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.mtfhip_batch_inline_warp.argtypes = [C.c_void_p]
        L.mtfhip_ssm_estimate_state_sigma.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
        L.mtfhip_pf_set_distributions.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mtfhip_pf_set_distr_draws.argtypes = [C.c_void_p, C.c_void_p]
        L.mtfhip_pf_get_distributions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mtfhip_timing_get_busy.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def check(code):
    if code != 0:
        msg = lib().mtfhip_last_error().decode("utf-8", "replace")
        raise _ERR.get(code, MtfHipError)(code, msg)
