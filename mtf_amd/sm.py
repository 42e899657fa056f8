"""Search methods that drive the device hot path: the callers of SURVEY.md section 8a rows A12 / A13.

ESM / FCLK / ICLK mirror nt::ESM / nt::FCLK / nt::ICLK (SM/src/NT/{ESM,FCLK,ICLK}.cc): `initialize`,
`update`, `set_region`, `get_region` with the reference's parameter names and defaults.  Two execution
modes: host_solve=True keeps the S x S solve, Levenberg-Marquardt and the compositional update on the
host exactly as the reference does (one fused launch + one small read-back per iteration);
host_solve=False runs the whole loop on the device (mtfhip_batch_track).

PF mirrors SM/src/PF.cc with the per-particle scoring on the device and, optionally, sharded over GPUs.
"""
import os

import numpy as np

from . import _lib as L
from .api import Batch, sm_desc


def _solve(H, g):
    """state_update = -H.colPivHouseholderQr().solve(g^T): any backward-stable solver agrees to ~1e-12
    on these well-conditioned S x S systems; symmetric diagonal scaling keeps it so for the badly scaled
    homography Hessian."""
    d = np.sqrt(np.abs(np.diag(H)))
    d[d == 0] = 1.0
    return -np.linalg.solve(H / np.outer(d, d), g / d) / d


class LKTracker:
    """One or many (B) independent targets tracked with ESM / FCLK / ICLK + SSD, NCC or MI on one GPU through the fused path
    (mtfhip_batch_iterate: k_fused_ssd / k_fused_ncc, or the fused MI passes): host_solve=True = fused launch(es) per
    iteration + the reference's pivoted QR and Levenberg-Marquardt on the host; False = the whole loop on the device
    (mtfhip_batch_track)."""

    def __init__(self, ctx, sm, ssm=L.SSM_HOMOGRAPHY, resx=50, resy=50, n_targets=1, host_solve=True, am=L.AM_SSD,
                 am_params=None, **params):
        self.ctx = ctx
        self.batch = Batch(ctx, am, ssm, resx, resy, n_targets, **(am_params or {}))   # e.g. n_channels=3: MCSSD / MCNCC
        self.B, self.S = n_targets, self.batch.S
        self.host_solve = host_solve
        self.sm = sm_desc(sm, **params)
        self.n_iters = np.zeros(n_targets, dtype=np.int32)

    # nt::*::initialize (NT/ESM.cc:110-146, NT/FCLK.cc:102-169, NT/ICLK.cc:71-128)
    def initialize(self, corners):
        self.batch.set_corners(np.asarray(corners, dtype=np.float64).reshape(self.B, 2, 4))
        self.batch.init_template(self.sm)

    # nt::*::setRegion (NT/ESM.cc:148-168, NT/FCLK.cc:360-376, NT/ICLK.cc:131-157): SSM reset, template kept, ESM's J0 / H0 refreshed
    def set_region(self, corners):
        self.batch.set_region(np.asarray(corners, dtype=np.float64).reshape(self.B, 2, 4), self.sm)

    def get_region(self):
        return self.batch.get_corners()

    def update_region(self, corners):
        """set_region(corners) followed by update(): the pair GridTracker and PyramidalTracker issue per frame, as one
        C-ABI call on the device-loop path (mtfhip_batch_track_region)"""
        if self.host_solve:
            self.set_region(corners)
            return self.update()
        self.n_iters, out = self.batch.track_region(np.asarray(corners, dtype=np.float64).reshape(self.B, 2, 4), self.sm)
        return out

    def update(self):
        if not self.host_solve:
            self.n_iters, corners = self.batch.track(self.sm)
            return corners
        sm, b = self.sm, self.batch
        B = self.B
        active = np.ones(B, dtype=bool)
        prev_f = np.zeros(B)
        delta = np.full(B, sm.lm_delta_init)
        last_dp = np.zeros((B, self.S))
        self.n_iters[:] = 0
        # the accept/reject state machine of NT/FCLK.cc:193-217, vectorised over targets.  ESM (NT/ESM.cc:179,224) and ICLK
        # (NT/ICLK.cc:169,187) are `for` loops whose `continue` still advances iter_id: there a rejected pass consumes an iteration
        it = 0
        state_reset = np.zeros(B, dtype=bool)
        while it < sm.max_iters and active.any():
            f, g, H = b.iterate(sm)
            dps = np.zeros((B, self.S))
            undo = np.zeros(B, dtype=bool)
            if sm.leven_marq:
                for t in range(B):
                    if not active[t] or state_reset[t]:
                        continue
                    if self.n_iters[t] > 0:
                        if f[t] < prev_f[t]:
                            delta[t] *= sm.lm_delta_update
                            undo[t] = True
                            continue
                        if f[t] > prev_f[t]:
                            delta[t] /= sm.lm_delta_update
                    prev_f[t] = f[t]
            prev_corners = b.get_corners()
            if not sm.leven_marq:
                # no accept / reject bookkeeping: all targets solved in one batched call (same scaling as _solve)
                d = np.sqrt(np.abs(np.diagonal(H, axis1=1, axis2=2)))
                d[d == 0] = 1.0
                dp_all = -np.linalg.solve(H / (d[:, :, None] * d[:, None, :]), (g / d)[..., None])[..., 0] / d
                dp_all[~active] = 0
                last_dp = dp_all
                dps = b.invert_state(dp_all) if sm.sm == L.SM_ICLK else dp_all.copy()
                dps[~active] = 0
            for t in range(B if sm.leven_marq else 0):
                if not active[t]:
                    continue
                if undo[t]:
                    # undo the last update: FCLK / ESM apply the inverse, ICLK re-applies the forward update
                    dps[t] = last_dp[t] if sm.sm == L.SM_ICLK else b.invert_state(np.tile(last_dp[t], (B, 1)))[0]
                    continue
                Ht = H[t].copy()
                if sm.leven_marq:
                    Ht[np.diag_indices(self.S)] += delta[t] * np.diag(Ht)
                dp = _solve(Ht, g[t])
                last_dp[t] = dp
                dps[t] = b.invert_state(np.tile(dp, (B, 1)))[0] if sm.sm == L.SM_ICLK else dp
            b.compositional_update(dps)
            corners = b.get_corners()
            change = ((prev_corners - corners) ** 2).reshape(B, -1).sum(axis=1)
            for t in range(B):
                if not active[t]:
                    continue
                if undo[t]:
                    state_reset[t] = True
                    if sm.sm != L.SM_FCLK:
                        self.n_iters[t] += 1
                        if self.n_iters[t] >= sm.max_iters:
                            active[t] = False
                    continue
                state_reset[t] = False
                self.n_iters[t] += 1
                if change[t] < sm.epsilon or self.n_iters[t] >= sm.max_iters:
                    active[t] = False
            it += 1 if not undo.any() else 0
            if it == 0 and undo.all():
                break
        return b.get_corners()


class NTSearchMethod:
    """nt::ESM / nt::FCLK / nt::ICLK written against the AM / SSM interface only, one C-ABI call per
    reference virtual, exactly in the reference's order (SM/src/NT/ESM.cc:170-296, NT/FCLK.cc:171-358,
    NT/ICLK.cc:160-299).  Works for every appearance model the device path implements (SSD, NCC, MI);
    this is the literal drop-in shape -- LKTracker is the fused fast path for SSD.  Levenberg-Marquardt
    is not vectorised here (use LKTracker(host_solve=True) for SSD)."""

    def __init__(self, ctx, sm, am=L.AM_SSD, ssm=L.SSM_HOMOGRAPHY, resx=50, resy=50, n_targets=1, am_params=None,
                 **params):
        """am_params may carry n_channels=3: MCSSD / MCNCC / MCMI (the context then holds an H x W x 3 float32 frame)"""
        self.batch = Batch(ctx, am, ssm, resx, resy, n_targets, **(am_params or {}))
        self.B, self.S = n_targets, self.batch.S
        self.sm = sm_desc(sm, **params)
        if self.sm.leven_marq:
            raise L.FunctionNotImplemented(-2, "NTSearchMethod: leven_marq is not vectorised")
        self.H0 = None
        self.trace = []

    def _pix_jacobian(self, init):
        b, sm = self.batch, self.sm
        grad, dst = (L.BUF_DI0_DX, L.BUF_J0) if init else (L.BUF_DIT_DX, L.BUF_JT)
        if sm.chained_warp:
            (b.initialize_pix_grad if init else b.update_pix_grad)()
            b.cmpt_pix_jacobian(L.JAC_WARPED, grad, dst)
        else:
            b.update_grad_pts()
            (b.initialize_pix_grad if init else b.update_pix_grad)(warped=True)
            b.cmpt_pix_jacobian(L.JAC_INIT, grad, dst)

    # second order (sec_ord_hess): ESM::initializePixHessian / updatePixHessian NT/ESM.cc:406-432, inlined by
    # FCLK (NT/FCLK.cc:121-143,243-257) and ICLK (NT/ICLK.cc:96-113,223-237)
    def _pix_hess(self, init):
        b = self.batch
        fn = b.initialize_pix_hess if init else b.update_pix_hess
        if self.sm.chained_warp:
            fn()
        else:
            b.update_hess_pts()
            fn(warped=True)

    def _pix_hessian(self, init):
        b = self.batch
        hess, grad, dst = ((L.BUF_D2I0_DX2, L.BUF_DI0_DX, L.BUF_D2I0_DP2) if init else
                           (L.BUF_D2IT_DX2, L.BUF_DIT_DX, L.BUF_D2IT_DP2))
        b.cmpt_pix_hessian(L.JAC_WARPED if self.sm.chained_warp else L.JAC_INIT, hess, grad, dst)

    def _self_hessian(self, j, d2):
        b = self.batch
        return b.cmpt_self_hessian2(j, d2) if self.sm.sec_ord_hess else b.cmpt_self_hessian(j)

    def initialize(self, corners):
        b, sm = self.batch, self.sm
        so = bool(sm.sec_ord_hess)
        b.set_corners(np.asarray(corners, dtype=np.float64).reshape(self.B, 2, 4))
        b.initialize_pix_vals()
        if sm.sm == L.SM_ESM:
            self._pix_jacobian(True)
            if so:
                self._pix_hess(True); self._pix_hessian(True)
            b.initialize_similarity(); b.initialize_grad(); b.initialize_hess()
            if sm.hess_type in (0, 2):
                self.H0 = self._self_hessian(L.BUF_J0, L.BUF_D2I0_DP2)
        elif sm.sm == L.SM_FCLK:
            b.initialize_similarity(); b.initialize_grad(); b.initialize_hess()
            self._pix_jacobian(True)
            if so:
                self._pix_hess(True)
            if sm.hess_type == 0:
                if so:
                    self._pix_hessian(True)
                self.H0 = self._self_hessian(L.BUF_J0, L.BUF_D2I0_DP2)
        else:
            self._pix_jacobian(True)
            b.initialize_similarity(); b.initialize_grad(); b.initialize_hess()
            b.cmpt_init_jacobian(L.BUF_J0)
            if so:
                self._pix_hess(True)
                if sm.hess_type != 1:
                    self._pix_hessian(True)
            if sm.hess_type == 0:
                self.H0 = self._self_hessian(L.BUF_J0, L.BUF_D2I0_DP2)

    def get_region(self):
        return self.batch.get_corners()

    def _esm_iter(self):
        b, sm = self.batch, self.sm
        b.update_pix_vals()
        b.update_similarity(False)
        self._pix_jacobian(False)
        if sm.jac_type == 0 or sm.hess_type == 3:
            b.mean_jacobian()
        so = bool(sm.sec_ord_hess)
        if so and sm.hess_type != 0:
            self._pix_hess(False); self._pix_hessian(False)
        b.update_curr_grad(); b.update_init_grad()
        g = b.cmpt_curr_jacobian(L.BUF_JM) if sm.jac_type == 0 else 0.5 * b.cmpt_difference_of_jacobians()
        ht = sm.hess_type
        if ht == 0:
            H = self.H0
        elif ht == 3:
            if so:
                b.mean_pix_hessian()
                H = b.cmpt_curr_hessian2(L.BUF_JM, L.BUF_D2IM_DP2)
            else:
                H = b.cmpt_curr_hessian(L.BUF_JM)
        elif ht == 4:
            H = 0.5 * (b.cmpt_sum_of_hessians2() if so else b.cmpt_sum_of_hessians())
        elif ht == 2:
            H = 0.5 * (self._self_hessian(L.BUF_JT, L.BUF_D2IT_DP2) + self.H0)
        elif ht == 1:
            H = self._self_hessian(L.BUF_JT, L.BUF_D2IT_DP2)
        else:
            H = b.cmpt_curr_hessian2() if so else b.cmpt_curr_hessian(L.BUF_JT)
        return g, H

    def _fclk_iter(self):
        b, sm = self.batch, self.sm
        b.update_pix_vals()
        b.update_similarity(False)
        b.update_curr_grad()
        self._pix_jacobian(False)
        so = bool(sm.sec_ord_hess)
        if so and sm.hess_type != 0:
            self._pix_hess(False); self._pix_hessian(False)
        g = b.cmpt_curr_jacobian(L.BUF_JT)
        if sm.hess_type == 0:
            H = self.H0
        elif sm.hess_type == 1:
            H = self._self_hessian(L.BUF_JT, L.BUF_D2IT_DP2)
        else:
            H = b.cmpt_curr_hessian2() if so else b.cmpt_curr_hessian(L.BUF_JT)
        return g, H

    def _iclk_iter(self):
        b, sm = self.batch, self.sm
        b.update_pix_vals()
        b.update_similarity(False)
        b.update_init_grad()
        g = b.cmpt_init_jacobian(L.BUF_J0)
        if sm.hess_type == 0:
            H = self.H0
        elif sm.hess_type == 1:
            self._pix_jacobian(False)
            if sm.sec_ord_hess:
                self._pix_hess(False); self._pix_hessian(False)
            H = self._self_hessian(L.BUF_JT, L.BUF_D2IT_DP2)
        else:
            H = b.cmpt_init_hessian2() if sm.sec_ord_hess else b.cmpt_init_hessian(L.BUF_J0)
        return g, H

    def update(self):
        b, sm = self.batch, self.sm
        step = {L.SM_ESM: self._esm_iter, L.SM_FCLK: self._fclk_iter, L.SM_ICLK: self._iclk_iter}[sm.sm]
        active = np.ones(self.B, dtype=bool)
        self.trace = []
        for _ in range(sm.max_iters):
            g, H = step()
            f = b.get_similarity()
            dps = np.zeros((self.B, self.S))
            for t in range(self.B):
                if active[t]:
                    dps[t] = _solve(H[t], g[t])
            self.trace.append(dict(f=f.copy(), g=g.copy(), H=H.copy(), dp=dps.copy()))
            prev = b.get_corners()
            upd = b.invert_state(dps) if sm.sm == L.SM_ICLK else dps
            upd[~active] = 0
            b.compositional_update(upd)
            change = ((prev - b.get_corners()) ** 2).reshape(self.B, -1).sum(axis=1)
            active &= ~(change < sm.epsilon)
            if not active.any():
                break
        return b.get_corners()


def least_squares_estimator(ssm):
    """An all-points least-squares fit of the grid SSM to the patch centroids, standing in for ssm.estimateWarpFromPts
    (SSM/src/Homography.cc:885-897, Affine.cc:359-369 -> utils::estimateHomography / estimateAffine: RANSAC / LMedS over the same
    point pairs, out of scope -- SURVEY.md section 2).  Returns f(prev_pts (n, 2), curr_pts (n, 2)) -> state update in the SSM's
    own parameterisation (homography [h00-1, h01, h02, h10, h11-1, h12, h20, h21]; affine [tx, ty, a-1, b, c, d-1])."""
    def fit(prev_pts, curr_pts):
        a, b = np.asarray(prev_pts, dtype=np.float64), np.asarray(curr_pts, dtype=np.float64)
        n = len(a)
        if ssm == L.SSM_AFFINE:
            A = np.hstack([a, np.ones((n, 1))])
            M = np.linalg.lstsq(A, b, rcond=None)[0].T          # 2 x 3
            return np.array([M[0, 2], M[1, 2], M[0, 0] - 1, M[0, 1], M[1, 0], M[1, 1] - 1])
        # normalised DLT (Hartley): both point sets to zero mean and mean distance sqrt(2)
        def norm(p):
            m = p.mean(axis=0)
            sc = np.sqrt(2.0) / max(np.sqrt(((p - m) ** 2).sum(axis=1)).mean(), 1e-300)
            T = np.array([[sc, 0, -sc * m[0]], [0, sc, -sc * m[1]], [0, 0, 1.0]])
            return (p - m) * sc, T
        an, Ta = norm(a)
        bn, Tb = norm(b)
        rows = []
        for (x, y), (u, v) in zip(an, bn):
            rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
            rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
        h = np.linalg.svd(np.asarray(rows))[2][-1].reshape(3, 3)
        H = np.linalg.inv(Tb) @ h @ Ta
        H = H / H[2, 2]
        return np.array([H[0, 0] - 1, H[0, 1], H[0, 2], H[1, 0], H[1, 1] - 1, H[1, 2], H[2, 0], H[2, 1]])
    return fit


class GridTracker:
    """GridTracker<SSM> (SM/src/GridTracker.cc): grid_size x grid_size independent patch trackers laid over the tracked region by a
    grid SSM, all patches of a frame in ONE kernel launch (ICLK with a constant Hessian; other search methods take the loop of
    mtfhip_batch_track).  Parameter names, defaults and modes are GridTrackerParams' (GridTracker.cc:20-94, Config/parameters.h:505-510):
    patch_centroid_inside = 1 (default): the grid SSM has (grid_size + 1)^2 points and a patch is the patch_size rectangle centred on
    the centroid of its four surrounding grid points; 0: grid_size^2 points, patches centred on them; dyn_patch_size = 1: a patch is
    the quadrilateral of its four surrounding points.  reset_at_each_frame: 0 the patch trackers run on, 1 they are re-initialised
    on the new grid after every frame, 2 (any other value) only setRegion().  The grid points are ssm.getPts() of the grid SSM
    (the unit-square grid through the 4-corner homography, ProjectiveBase.cc:20-36) -- mtfhip_grid_layout.
    fb_err_thresh > 0 (shipped Config/modules.cfg:81: 2) switches the forward-backward error estimation on (:186-190, 294-343): after the
    frame's update every patch tracker -- re-initialised at its tracked location when fb_reinit (shipped 1) -- runs on the PREVIOUS frame,
    and the patches whose round trip misses their starting centroid by more than the threshold are left out of the fit (filled up to
    n_model_pts = est_params.n_model_pts, shipped 4).
    The robust fit of the grid SSM to the patch centroids (estimateWarpFromPts: RANSAC / LMedS) is out of scope (SURVEY.md section
    2): `estimator(prev_pts, curr_pts) -> state update` is pluggable, default an all-points least-squares fit."""

    def __init__(self, ctx, grid_size=10, patch_size=10, am=L.AM_NCC, ssm=L.SSM_AFFINE, max_iters=30, epsilon=1e-4, sm=L.SM_ICLK,
                 reset_at_each_frame=1, dyn_patch_size=0, patch_centroid_inside=1, grid_ssm=L.SSM_HOMOGRAPHY, estimator=None,
                 grid_size_y=None, patch_size_y=None, fb_err_thresh=0.0, fb_reinit=1, n_model_pts=4, **sm_params):
        self.ctx = ctx
        self.fb = L.GridFbDesc(float(fb_err_thresh), int(bool(fb_reinit)), int(n_model_pts)) if fb_err_thresh > 0 else None   # enable_fb_err_est :186-190
        self.fb_prev_pts = self.fb_err_mask = None
        self.grid_size, self.patch_size = grid_size, patch_size
        self.gd = L.GridDesc(grid_size, grid_size_y or grid_size, patch_size, patch_size_y or patch_size, int(reset_at_each_frame),
                             int(bool(dyn_patch_size)), int(bool(patch_centroid_inside)))
        self.n = self.gd.grid_size_x * self.gd.grid_size_y
        self.grid_ssm = grid_ssm
        self.estimator = estimator if estimator is not None else least_squares_estimator(grid_ssm)
        sm_params.setdefault("hess_type", 0)
        sm_params.setdefault("materialize", 0)
        self.tracker = LKTracker(ctx, sm, ssm, self.gd.patch_size_x, self.gd.patch_size_y, self.n, host_solve=False, am=am,
                                 max_iters=max_iters, epsilon=epsilon, **sm_params)
        self.region = None
        self._pending_region = None      # reset_at_each_frame = 2: the setRegion of the patch trackers rides in the next frame's launch
        self.prev_pts = np.zeros((self.n, 2), dtype=np.float32)
        self.curr_pts = np.zeros((self.n, 2), dtype=np.float32)
        self.ssm_update = np.zeros(8 if grid_ssm == L.SSM_HOMOGRAPHY else 6)

    def res(self):
        """GridTrackerParams::updateRes: the sampling resolution of the grid SSM"""
        import ctypes as C
        rx, ry = C.c_int(), C.c_int()
        L.check(L.lib().mtfhip_grid_res(C.byref(self.gd), C.byref(rx), C.byref(ry)))
        return rx.value, ry.value

    def _layout(self, region_corners, want_pts=False):
        import ctypes as C
        r = np.ascontiguousarray(np.asarray(region_corners, dtype=np.float64).reshape(2, 4).T)
        rx, ry = self.res()
        pts = np.empty((rx * ry, 2)) if want_pts else None
        pcs = np.empty((self.n, 4, 2))
        L.check(L.lib().mtfhip_grid_layout(C.byref(self.gd), r.ctypes.data, pts.ctypes.data if want_pts else None, pcs.ctypes.data))
        return pts, pcs.transpose(0, 2, 1).copy()

    def grid_pts(self, region_corners):
        """ssm.getPts() of the grid SSM laid over the region: (resx * resy, 2), row-major"""
        return self._layout(region_corners, True)[0]

    def patch_corners(self, region_corners):
        """the corners GridTracker::resetTrackers (:345-380) hands the patch trackers: (n, 2, 4)"""
        return self._layout(region_corners)[1]

    # GridTracker::initialize :233-246
    def initialize(self, region_corners):
        self.region = np.asarray(region_corners, dtype=np.float64).reshape(2, 4).copy()
        _, pp = self.tracker.batch.grid_reset(self.gd, self.tracker.sm, self.region, True)
        self.prev_pts[...] = pp
        self.curr_pts[...] = pp
        self._pending_region = None
        if self.fb is not None:
            self.ctx.keep_prev()         # prev_img = curr_img.clone() :241-243

    # GridTracker::setRegion :287-292
    def set_region(self, region_corners):
        self.region = np.asarray(region_corners, dtype=np.float64).reshape(2, 4).copy()
        self._reset()

    def get_region(self):
        return self.region.copy()

    def _reset(self):
        """resetTrackers(reinit_at_each_frame) :345-392"""
        if self.gd.reset_at_each_frame == 1:
            _, pp = self.tracker.batch.grid_reset(self.gd, self.tracker.sm, self.region, True)
            self.prev_pts[...] = pp
            self._pending_region = None
        else:
            # setRegion only: it rides in the next frame's launch (mtfhip_grid_frame with a region); the centroids are the patches'
            pcs = self.patch_corners(self.region)
            self.prev_pts[...] = (pcs.sum(axis=2) / 4.0).astype(np.float32)
            self._pending_region = self.region

    # GridTracker::update :247-285
    def update(self):
        """one frame: every patch tracker's update() (one launch), the fit of the grid SSM to prev_pts -> curr_pts, the region warped by
        it, and the reset the parameters ask for.  Returns the region's corners (2 x 4)."""
        if self.region is None:
            raise RuntimeError("GridTracker.update before initialize")
        if self.fb is not None:
            # :263-266 backwardEstimation(); prev_img = curr_img.clone()
            r = self.tracker.batch.grid_frame_fb(self.gd, self.tracker.sm, self.fb, self.prev_pts, self._pending_region)
            self._pending_region = None
            self.tracker.n_iters = r["n_iters"].copy()
            self.curr_pts[...] = r["centroids"]
            self.fb_prev_pts, self.fb_err_mask = r["fb_prev_pts"], r["fb_err_mask"]
            self.ctx.keep_prev()
            upd = np.asarray(self.estimator(r["prev_masked"].astype(np.float64), r["curr_masked"].astype(np.float64)), dtype=np.float64)
        else:
            n, _, cen = self.tracker.batch.grid_frame(self.gd, self.tracker.sm, self._pending_region)
            self._pending_region = None
            self.tracker.n_iters = n.copy()
            self.curr_pts[...] = cen
            upd = np.asarray(self.estimator(self.prev_pts.astype(np.float64), self.curr_pts.astype(np.float64)), dtype=np.float64)
        self.ssm_update = upd
        from .api import apply_warp_to_pts
        # ssm.applyWarpToCorners(opt_warped_corners, ssm.getCorners(), ssm_update); ssm.setCorners(opt_warped_corners) :270-272
        self.region = apply_warp_to_pts(self.grid_ssm, self.region, upd)
        if self.gd.reset_at_each_frame:
            self._reset()
        else:
            self.prev_pts[...] = self.curr_pts
        return self.region.copy()

    def update_patches(self, region_corners=None):
        """The patch half of a frame alone (GridTracker.cc:254-261), for callers that own the region: with region_corners (2 x 4, or
        the (n, 2, 4) patch corners themselves) the patch trackers are first reset to the grid laid over that region (setRegion),
        in the same C-ABI call and -- for ICLK -- the same launch.  Returns the patches' corners (n, 2, 4) and centroids (n, 2)."""
        if region_corners is None:
            corners = self.tracker.update()
            return corners, np.add.reduce(corners, axis=2) * 0.25   # utils::getCentroid miscUtils.h:473-480 (mean of the four corners)
        pc = region_corners if np.ndim(region_corners) == 3 else self.patch_corners(region_corners)
        if self.tracker.host_solve:
            corners = self.tracker.update_region(pc)
            return corners, np.add.reduce(corners, axis=2) * 0.25
        # one C-ABI call for the frame: reset + update of every patch, layout conversion and centroids included (mtfhip_grid_update)
        n, corners, centroids = self.tracker.batch.grid_update(pc, self.tracker.sm)
        self.tracker.n_iters = n.copy()
        return corners.copy(), centroids.copy()

    @property
    def n_iters(self):
        return self.tracker.n_iters


class Comm:
    """A communicator of the C-ABI collective (mtfhip_comm_*: RCCL directly, no torch).  rank 0 calls Comm.unique_id() and hands
    the 128 bytes to the other ranks over whatever channel the host program has; `torch_bootstrap` does it with a
    torch.distributed object broadcast when a process group exists (the bench and the tests use that)."""

    def __init__(self, rank=0, world=1, device=0, unique_id=None):
        import ctypes as C
        self._h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        L.check(L.lib().mtfhip_comm_create(buf, int(rank), int(world), int(device), C.byref(self._h)))
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id():
        import ctypes as C
        buf = (C.c_char * 128)()
        L.check(L.lib().mtfhip_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def loopback(cls, world, device=0):
        """`world` ranks as threads of this process on one device (mtfhip_comm_create_loopback): the sharded code path with the
        exchange done by a rendezvous of the threads -- one Comm per rank, each used from its own thread"""
        import ctypes as C
        arr = (C.c_void_p * world)()
        L.check(L.lib().mtfhip_comm_create_loopback(int(world), int(device), arr))
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c._h, c.rank, c.world = C.c_void_p(arr[r]), r, world
            out.append(c)
        return out

    @classmethod
    def detached(cls, rank, world, device=0):
        """rank and world with no collective behind them (mtfhip_comm_create_detached): a filter sharded over it exchanges its
        weights through peer stores, and the host program moves the mailbox handles (ParticleFilter(exchange_transport=...))"""
        import ctypes as C
        c = cls.__new__(cls)
        c._h = C.c_void_p()
        L.check(L.lib().mtfhip_comm_create_detached(int(rank), int(world), int(device), C.byref(c._h)))
        c.rank, c.world, c.is_detached = rank, world, world > 1
        return c

    @staticmethod
    def shard_bounds(n, world, rank):
        """(lo, count, per_rank) of mtfhip_pf_shard_bounds: the block a rank scores, per_rank = ceil(n / world)"""
        import ctypes as C
        lo, cnt, m = C.c_int(), C.c_int(), C.c_int()
        L.check(L.lib().mtfhip_pf_shard_bounds(int(n), int(world), int(rank), C.byref(lo), C.byref(cnt), C.byref(m)))
        return lo.value, cnt.value, m.value

    @classmethod
    def torch_bootstrap(cls, device):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return cls(0, 1, device)
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(rank, world, device, box[0])

    def world_as_seen(self):
        """the communicator's own rank count (mtfhip_comm_world)"""
        return int(L.lib().mtfhip_comm_world(self._h))

    def allgather(self, dev_send, count, dev_recv, stream=None):
        import ctypes as C
        L.check(L.lib().mtfhip_allgather_scores(self._h, C.c_void_p(dev_send), int(count), C.c_void_p(dev_recv),
                                                C.c_void_p(stream) if stream else None))

    def close(self):
        if self._h:
            L.lib().mtfhip_comm_destroy(self._h)
            self._h = None


class ParticleFilter:
    """nt::PF (SM/src/NT/PF.cc) over the device filter of the C ABI (mtfhip_pf_*): sample generation, scoring, cumulative
    weights, multinomial resampling and the estimate all run on the device; parameter names and enum values are the
    reference's (PFParams.h).  `comm` shards the scoring over the ranks of a Comm (one RCCL all-gather per iteration)."""

    def __init__(self, ctx, ssm=L.SSM_HOMOGRAPHY, resx=50, resy=50, n_particles=500,
                 ssm_sigma=(0.01, 0.01, 2.0, 0.01, 0.01, 2.0, 1e-5, 1e-5), ssm_mean=(0.0,) * 8, likelihood_alpha=1.0,
                 max_iters=1, epsilon=0.01, seed=0, am=L.AM_SSD, dynamic_model=0, update_type=1, likelihood_func=0,
                 resampling_type=1, mean_type=0, corner_based_sampling=0, reset_to_mean=0, measurement_sigma=0.1, ar_coeff=0.5,
                 comm=None, pt_based_sampling=0, n_channels=1, adaptive_resampling_thresh=0.0, update_distr_wts=0, min_distr_wt=0.1,
                 jacobian_as_sigma=0, pix_sigma=None, exchange="collective", exchange_transport=None):
        """pix_sigma: one value per sampler distribution; with pix_sigma[0] > 0 the sampler sigmas are estimated from them at
        initialize() (PFParams::processDistributions PFParams.cc:105-116, PF.cc:142-149: SSM::estimateStateSigma) and ssm_sigma is ignored.
        ssm_sigma / ssm_mean: one row of up to 8 values, or several rows = several sampler distributions (PFParams::processDistributions:
        the shipped Config/modules.cfg:157 uses five) whose weights follow the average particle weight each produced (update_distr_wts,
        min_distr_wt: PF.cc:345-369 -- update_distr_wts defaults to the reference's 0, and several distributions WITHOUT it are refused by
        every front end: the reference then draws from an all-zero discrete distribution, NT/PF.cc:241-257); adaptive_resampling_thresh in (0, 1]: resample only when the effective particle count drops to
        thresh * n (PF.cc:381-390); jacobian_as_sigma: the sampler's sigma of every frame is the Gauss-Newton step -H0^-1 g (PF.cc:58-64,
        156-165, 214-227)"""
        import ctypes as C
        # seed 0 = "draw one" (below) -- but only an unsharded filter may: the ranks of a sharded one must propose identical particles
        # (each scores a block of ITS proposals, the all-gather mixes the weights), so there seed 0 is refused before anything is
        # created (mtfhip_pf_set_comm cross-checks the seeds of all ranks as well)
        if not seed and comm is not None and comm.world > 1:
            raise ValueError("ParticleFilter(comm=...) over %d ranks needs an explicit non-zero seed shared by every rank" % comm.world)
        rows_s = [list(ssm_sigma)] if np.ndim(ssm_sigma) == 1 else [list(r) for r in ssm_sigma]
        rows_m = [list(ssm_mean)] if np.ndim(ssm_mean) == 1 else [list(r) for r in ssm_mean]
        self.pix_sigma = None
        if pix_sigma is not None and len(np.atleast_1d(pix_sigma)) and float(np.atleast_1d(pix_sigma)[0]) > 0:
            self.pix_sigma = [float(v) for v in np.atleast_1d(pix_sigma)]
            rows_s = [[1.0] * 8 for _ in self.pix_sigma]       # placeholders until initialize() knows the points
            rows_m = [[0.0] * 8 for _ in self.pix_sigma]
        if jacobian_as_sigma:      # n_distr = 1 (PF.cc:62)
            rows_s, rows_m = rows_s[:1], rows_m[:1]
        n_distr = max(len(rows_s), len(rows_m))
        while len(rows_s) < n_distr: rows_s.append(rows_s[-1])     # (PFParams.cc:131-166: the last sigma / mean row is reused)
        while len(rows_m) < n_distr: rows_m.append(rows_m[-1])
        ssm_sigma, ssm_mean = rows_s[0], rows_m[0]
        self.jacobian_as_sigma = bool(jacobian_as_sigma)
        # n_channels = 3: MCSSD / MCNCC (the context then holds an H x W x 3 float32 frame)
        self.batch = Batch(ctx, am, ssm, resx, resy, 1, likelihood_alpha=likelihood_alpha, n_channels=n_channels)
        self.S, self.n = self.batch.S, n_particles
        self.desc = L.PFDesc(n_particles, max_iters, epsilon, dynamic_model, update_type, likelihood_func, resampling_type, mean_type,
                             corner_based_sampling, reset_to_mean, measurement_sigma, ar_coeff)
        for k in range(8):
            self.desc.ssm_sigma[k] = float(ssm_sigma[k]) if k < len(ssm_sigma) else 0.0
            self.desc.ssm_mean[k] = float(ssm_mean[k]) if k < len(ssm_mean) else 0.0
        # the reference seeds its generators from random_device (PF.cc:97-105): seed 0 = "draw one", anything else is reproducible
        self.desc.seed = int(seed) if seed else int.from_bytes(os.urandom(8), "little") | 1
        self.desc.pt_based_sampling = int(pt_based_sampling)
        self.desc.adaptive_resampling_thresh = float(adaptive_resampling_thresh)
        self.desc.update_distr_wts = int(update_distr_wts) if n_distr > 1 else 0     # (PF.cc:67)
        self.desc.min_distr_wt = float(min_distr_wt)
        self.n_distr = n_distr
        self._h = C.c_void_p()
        rc = L.lib().mtfhip_pf_create(self.batch._h, C.byref(self.desc), C.byref(self._h))
        if rc != 0:
            self._h = None
            self.batch.close()
            L.check(rc)
        ctx._dependents.add(self)
        if n_distr > 1:
            sg, mn = np.zeros((n_distr, 8)), np.zeros((n_distr, 8))
            for i in range(n_distr):
                sg[i, :min(8, len(rows_s[i]))] = rows_s[i][:8]; mn[i, :min(8, len(rows_m[i]))] = rows_m[i][:8]
            L.check(L.lib().mtfhip_pf_set_distributions(self._h, n_distr, sg.ctypes.data_as(C.c_void_p), mn.ctypes.data_as(C.c_void_p)))
        self.comm = comm
        if comm is not None:
            L.check(L.lib().mtfhip_pf_set_comm(self._h, comm._h))
        # how the weights of a sharded filter travel: "collective" (one RCCL all-gather per iteration, the default) or "peer" (the
        # scoring kernel stores into every rank's mailbox, the scan waits for arrival counters: mtfhip_pf_set_exchange)
        if exchange not in ("collective", "peer"):
            raise ValueError("exchange must be 'collective' or 'peer', not %r" % (exchange,))
        if exchange == "peer" and exchange_transport is not None and comm is None:
            raise ValueError("exchange='peer' with an exchange_transport belongs to a filter sharded over a communicator (comm=...)")
        self.exchange = exchange
        if exchange == "peer" and exchange_transport is not None:
            # the host program moves the 64-byte mailbox handles: exchange_transport(mine: bytes) -> [every rank's bytes, rank-major]
            buf = (C.c_char * 64)()
            L.check(L.lib().mtfhip_pf_exchange_export(self._h, buf))
            handles = list(exchange_transport(bytes(buf)))
            if len(handles) != comm.world or any(len(h) != 64 for h in handles):
                raise ValueError("exchange_transport must return the %d ranks' 64-byte handles" % comm.world)
            L.check(L.lib().mtfhip_pf_exchange_connect(self._h, (C.c_char * (64 * comm.world)).from_buffer_copy(b"".join(handles))))
        elif exchange == "peer":
            L.check(L.lib().mtfhip_pf_set_exchange(self._h, 1))
        elif comm is not None and getattr(comm, "is_detached", False):
            raise ValueError("a detached communicator has no collective: exchange='peer' with an exchange_transport")
        if ssm == L.SSM_HOMOGRAPHY:
            self.nz = 10 if corner_based_sampling else 8
        else:
            self.nz = 8 if pt_based_sampling == 2 else 6
        self.n_iters = 0

    def close(self):
        if getattr(self, "_h", None):
            L.lib().mtfhip_pf_destroy(self._h)
            self._h = None
        if getattr(self, "batch", None) is not None:
            self.batch.close()

    def __del__(self):   # a filter dropped without close() must not keep its device buffers until the Context goes
        try:
            self.close()
        except Exception:
            pass

    def set_max_similarity(self, f):
        """max_similarity = am->getSimilarity() after am->updateModel (PF.cc:443-446)"""
        L.check(L.lib().mtfhip_pf_set_max_similarity(self._h, L.C.c_double(float(f))))

    # nt::PF::initialize (NT/PF.cc:136-183)
    def initialize(self, corners):
        self.batch.set_corners(np.asarray(corners, dtype=np.float64).reshape(1, 2, 4))
        self.batch.initialize_pix_vals()
        self.batch.initialize_similarity()
        if self.pix_sigma is not None:   # PF.cc:142-149: one estimated sigma row per distribution, zero means
            import ctypes as C
            sg = np.zeros((self.n_distr, 8)); mn = np.zeros((self.n_distr, 8))
            for i in range(self.n_distr):
                sg[i, :self.S] = self.batch.estimate_state_sigma(self.pix_sigma[i])[0]
            if self.n_distr > 1:
                L.check(L.lib().mtfhip_pf_set_distributions(self._h, self.n_distr, sg.ctypes.data_as(C.c_void_p), mn.ctypes.data_as(C.c_void_p)))
            else:
                L.check(L.lib().mtfhip_pf_set_sampler(self._h, sg[0].ctypes.data_as(C.c_void_p), mn[0].ctypes.data_as(C.c_void_p)))
            self.state_sigma = sg
        if self.jacobian_as_sigma:   # PF.cc:156-165: d2f_dp2 = the self Hessian of the template's pixel Jacobian
            b, additive = self.batch, self.desc.update_type == 0
            b.initialize_grad()
            b.initialize_pix_grad()
            b.cmpt_pix_jacobian(L.JAC_PIX if additive else L.JAC_WARPED, L.BUF_DI0_DX, L.BUF_J0)
            self._d2f_dp2 = b.cmpt_self_hessian(L.BUF_J0)[0]
        L.check(L.lib().mtfhip_pf_initialize(self._h))

    def _jacobian_sigma(self):
        """PF.cc:214-227: state_sigma[0] = -d2f_dp2.colPivHouseholderQr().solve(df_dp^T) on the current frame, then setSampler"""
        import ctypes as C
        b, additive = self.batch, self.desc.update_type == 0
        b.update_pix_vals(); b.update_similarity(False); b.update_curr_grad(); b.update_pix_grad()
        b.cmpt_pix_jacobian(L.JAC_PIX if additive else L.JAC_WARPED, L.BUF_DIT_DX, L.BUF_JT)
        g = b.cmpt_curr_jacobian(L.BUF_JT)[0]
        sigma = np.zeros(8); sigma[:self.S] = -np.linalg.solve(self._d2f_dp2, g)
        mean = np.array([self.desc.ssm_mean[k] for k in range(8)])
        L.check(L.lib().mtfhip_pf_set_sampler(self._h, sigma.ctypes.data_as(C.c_void_p), mean.ctypes.data_as(C.c_void_p)))
        return sigma

    def distributions(self):
        """(distribution weights the next iteration draws from, distribution id of every particle's CURRENT proposal -- with look-ahead
        proposals, the default with the device generator, those are the pending iteration's draws (mtfhip.h) --, whether the last
        iteration resampled)"""
        import ctypes as C
        w, ids, res = np.ones(self.n_distr), np.zeros(self.n, dtype=np.int32), C.c_int(1)
        L.check(L.lib().mtfhip_pf_get_distributions(self._h, w.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), C.byref(res)))
        return w, ids, bool(res.value)

    def set_region(self, corners):
        c = self.batch._corners_in(np.asarray(corners, dtype=np.float64).reshape(1, 2, 4))
        L.check(L.lib().mtfhip_pf_set_region(self._h, c.ctypes.data_as(L.C.c_void_p)))

    def get_region(self):
        return self.batch.get_corners()

    def iteration(self, normals=None, uniforms=None, distr_uniforms=None):
        """one iteration of update()'s loop; normals (n, nz) / uniforms (n,) / distr_uniforms (n,: the distribution draws, several
        sampler distributions only) or None for the device generator"""
        import ctypes as C
        if distr_uniforms is not None:
            du = np.ascontiguousarray(np.asarray(distr_uniforms, dtype=np.float64).reshape(self.n))
            L.check(L.lib().mtfhip_pf_set_distr_draws(self._h, du.ctypes.data_as(C.c_void_p)))
        nz = None if normals is None else np.ascontiguousarray(np.asarray(normals, dtype=np.float64).reshape(self.n, self.nz))
        un = None if uniforms is None else np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64).reshape(self.n))
        norm = C.c_double()
        L.check(L.lib().mtfhip_pf_iteration(self._h, None if nz is None else nz.ctypes.data_as(C.c_void_p),
                                            None if un is None else un.ctypes.data_as(C.c_void_p), C.byref(norm)))
        return norm.value

    def update(self):
        if self.jacobian_as_sigma:
            self._jacobian_sigma()
        u = self.__dict__.get("_upd")
        if u is None:     # per-frame call: the out-parameter and the bound function are made once
            import ctypes as C
            n = C.c_int()
            u = self._upd = (n, C.byref(n), L.lib().mtfhip_pf_update)
        n, pn, fn = u
        L.check(fn(self._h, pn))
        self.n_iters = n.value
        return self.batch.get_corners()

    def particles(self):
        import ctypes as C
        st, ar, w = np.empty((self.n, self.S)), np.empty((self.n, self.S)), np.empty(self.n)
        ids = np.empty(self.n, dtype=np.int32)
        L.check(L.lib().mtfhip_pf_get_particles(self._h, st.ctypes.data_as(C.c_void_p), ar.ctypes.data_as(C.c_void_p),
                                                w.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p)))
        return st, ar, w, ids

    def set_particles(self, states, ars=None):
        import ctypes as C
        st = np.ascontiguousarray(np.asarray(states, dtype=np.float64).reshape(self.n, self.S))
        ar = None if ars is None else np.ascontiguousarray(np.asarray(ars, dtype=np.float64).reshape(self.n, self.S))
        L.check(L.lib().mtfhip_pf_set_particles(self._h, st.ctypes.data_as(C.c_void_p), None if ar is None else ar.ctypes.data_as(C.c_void_p)))

    @property
    def max_similarity(self):
        return L.lib().mtfhip_pf_max_similarity(self._h)

    @staticmethod
    def _warp(ssm, p):
        if ssm == L.SSM_HOMOGRAPHY:
            return np.array([[1 + p[0], p[1], p[2]], [p[3], 1 + p[4], p[5]], [p[6], p[7], 1.0]])
        return np.array([[1 + p[2], p[3], p[0]], [p[4], 1 + p[5], p[1]], [0, 0, 1.0]])

    def _compose(self, base, pert):
        """compositionalRandomWalk (Homography.cc:916-926, Affine analogue): W(base) * W(pert) -- host helper of NNDataset"""
        ssm = self.batch.desc.ssm
        out = np.empty_like(base)
        for k in range(base.shape[0]):
            W = ParticleFilter._warp(ssm, base[k]) @ ParticleFilter._warp(ssm, pert[k])
            if ssm == L.SSM_HOMOGRAPHY:
                W = W / W[2, 2]
                out[k] = [W[0, 0] - 1, W[0, 1], W[0, 2], W[1, 0], W[1, 1] - 1, W[1, 2], W[2, 0], W[2, 1]]
            else:
                out[k] = [W[0, 2], W[1, 2], W[0, 0] - 1, W[0, 1], W[1, 0], W[1, 1] - 1]
        return out

    @staticmethod
    def binary_multinomial_resample(wts, uniforms):
        """PF::binaryMultinomialResampling PF.cc:455-502 (host twin of k_pf_resample, used by the CPU tests): smallest index
        whose normalised cumulative weight is >= the uniform draw."""
        cum = np.cumsum(wts)
        cum = cum / cum[-1]
        return np.minimum(np.searchsorted(cum, uniforms, side="left"), len(wts) - 1)


class NNDataset:
    """Dataset generation of nt::NN (SM/src/NT/NN.cc:131-191, compositional update) in ONE launch (mtfhip_nn_dataset: perturbation draw or
    the caller's perturbations, invertState, compositionalUpdate, updatePixVals, updateDistFeat per sample -- SSD, NCC and MI features,
    single- and multi-channel).  Several sampler distributions (NN.cc:56-84: state_sigma[k], distr_n_samples[k]) are consecutive row
    blocks, one launch each.  With `group` (a torch.distributed process group) the rows are block-partitioned over its ranks and one
    all-gather leaves the whole matrix on every rank (SURVEY.md section 8e's partition; the draws are keyed by the global sample index).
    The index built over the matrix (FLANN / GNN in the reference, NN.cc:99-128) is outside the path -- `nearest` is the exhaustive search."""

    def __init__(self, ctx, am=L.AM_SSD, ssm=L.SSM_HOMOGRAPHY, resx=50, resy=50, n_samples=1000,
                 ssm_sigma=(0.01, 0.01, 2.0, 0.01, 0.01, 2.0, 1e-5, 1e-5), ssm_mean=None, seed=0, am_params=None, distr_n_samples=None):
        self.batch = Batch(ctx, am, ssm, resx, resy, 1, **(am_params or {}))
        self.S = self.batch.S
        self.n = int(n_samples)
        sg = np.atleast_2d(np.asarray(ssm_sigma, dtype=np.float64))
        self.sigmas = [row[: self.S] for row in sg]
        mn = np.zeros_like(sg) if ssm_mean is None else np.atleast_2d(np.asarray(ssm_mean, dtype=np.float64))
        self.means = [row[: self.S] for row in mn]
        # distr_n_samples (NN.cc:60-73): samples per distribution; default: equal shares, the remainder to the last one
        k = len(self.sigmas)
        if distr_n_samples is None:
            distr_n_samples = [self.n // k] * k
            distr_n_samples[-1] += self.n - sum(distr_n_samples)
        assert sum(distr_n_samples) == self.n and len(distr_n_samples) == k
        self.distr_n_samples = list(distr_n_samples)
        self.seed = int(seed)
        self.sigma = self.sigmas[0]
        self.rng = np.random.default_rng(seed)     # (for callers that make their own perturbations)
        self.perturbations = None
        self.features = None

    def feature_size(self):
        return self.batch.nn_feature_size()

    def initialize(self, corners, perturbations=None):
        """NN::initialize's dataset half (NT/NN.cc:85-113): template at `corners`, then generateDataset -> features (n, feat_size)"""
        b = self.batch
        b.set_corners(np.asarray(corners, dtype=np.float64).reshape(1, 2, 4))
        b.initialize_pix_vals()
        perts, feats, lo = [], [], 0
        for k, cnt in enumerate(self.distr_n_samples):
            if cnt == 0:
                continue
            pin = None if perturbations is None else np.asarray(perturbations, dtype=np.float64).reshape(self.n, self.S)[lo:lo + cnt]
            p, f = b.nn_dataset(cnt, self.sigmas[k], self.means[k], seed=self.seed + k, perturbations=pin)
            perts.append(p); feats.append(f); lo += cnt
        self.perturbations, self.features = np.concatenate(perts), np.concatenate(feats)
        return self.features

    def initialize_sharded(self, corners, group=None, device=None):
        """the same with the rows block-partitioned over the ranks of `group` (torch.distributed; backend nccl = RCCL on the GPUs of a node, gloo
        in the CPU tests' stand-in): rank r generates rows mtf_amd.dist.shard_bounds(n, r, world) into its slice of a device buffer and ONE
        all-gather completes the matrix on every rank.  Single distribution.  -> torch tensor (n, feat_size) on `device`."""
        import torch
        import torch.distributed as tdist
        from . import dist as mdist
        world = tdist.get_world_size(group) if tdist.is_initialized() else 1
        rank = tdist.get_rank(group) if tdist.is_initialized() else 0
        b = self.batch
        b.set_corners(np.asarray(corners, dtype=np.float64).reshape(1, 2, 4))
        b.initialize_pix_vals()
        F, n = self.feature_size(), self.n
        lo, cnt, m = mdist.padded_shard(n, rank, world)       # rows per rank of the padded buffer (the filter's partition: ceil(n / world))
        buf = torch.zeros((m * world, F), dtype=torch.float64, device=device)
        d = b.nn_desc(n, self.sigmas[0], self.means[0], self.seed)
        b.nn_dataset_dev(d, buf[rank * m:].data_ptr(), lo, cnt)
        b.ctx.synchronize()
        if world > 1:
            tdist.all_gather_into_tensor(buf, buf[rank * m:(rank + 1) * m].clone(), group=group)
        self.features = buf[:n]
        return self.features

    def nearest(self, feature):
        d = ((self.features - np.asarray(feature)[None]) ** 2).sum(axis=1)
        k = int(np.argmin(d))
        return k, float(d[k])


class PyramidalTracker:
    """PyramidalTracker (SM/src/PyramidalTracker.cc): one tracker per pyramid level, coarse to fine.  Each level owns
    a Context whose image is derived on the device from the level above (cv::pyrDown for scale_factor 0.5, else
    resize + GaussianBlur), so a frame is uploaded once.  `make_tracker(ctx, level)` builds the per-level tracker
    (anything with initialize / update / set_region / get_region over (1, 2, 4) corners, e.g. LKTracker)."""

    def __init__(self, ctx0, make_tracker, no_of_levels=3, scale_factor=0.5):
        from .api import Context
        self.n, self.scale = no_of_levels, float(scale_factor)
        self.ctxs = [ctx0] + [Context(ctx0.device) for _ in range(no_of_levels - 1)]
        self.trackers = [make_tracker(c, k) for k, c in enumerate(self.ctxs)]
        self.sizes = None
        self.overall = self.scale ** (no_of_levels - 1)

    def update_image_pyramid(self):
        """PyramidalTracker::updateImagePyramid :88-97 (level sizes as in setImage :50-53)"""
        if self.sizes is None:
            r, c = self.ctxs[0].image_shape()
            self.sizes = [(r, c)]
            for _ in range(1, self.n):
                r, c = int(r * self.scale), int(c * self.scale)
                self.sizes.append((r, c))
        for k in range(1, self.n):
            self.ctxs[k].pyramid_level_from(self.ctxs[k - 1], self.sizes[k][0], self.sizes[k][1], pyr_down=(self.scale == 0.5))

    def initialize(self, corners):
        self.update_image_pyramid()
        c = np.asarray(corners, dtype=np.float64).reshape(1, 2, 4).copy()
        self.trackers[0].initialize(c)
        for k in range(1, self.n):
            c = c * self.scale
            self.trackers[k].initialize(c)

    def update(self):
        """:117-131: coarsest level first, every finer level starts from the scaled-up result"""
        self.update_image_pyramid()
        self.trackers[-1].update()
        out = self.trackers[-1].get_region()
        for k in range(self.n - 2, -1, -1):
            t = self.trackers[k]
            if hasattr(t, "update_region"):     # setRegion + update of the level in one C-ABI call (mtfhip_batch_track_region)
                t.update_region(out / self.scale)
            else:
                t.set_region(out / self.scale)
                t.update()
            out = t.get_region()
        self.trackers[-1].set_region(out * self.overall)
        return out

    def set_region(self, corners):
        c = np.asarray(corners, dtype=np.float64).reshape(1, 2, 4).copy()
        self.trackers[0].set_region(c)
        for k in range(1, self.n):
            c = c * self.scale
            self.trackers[k].set_region(c)

    def get_region(self):
        return self.trackers[0].get_region()
