#!/usr/bin/env python
"""Generates tests/golden/lk_golden.npz from the independent NumPy re-derivation (oracle/numpy_ref.py).

The reference cannot run here (Eigen / OpenCV / Boost absent) and ships no golden vectors, so these
fixtures come from a second, independently written float64 implementation of the maths; they pin
the C++ oracle (tests/test_oracle_golden.py), which in turn is the checker for the HIP path.
PARITY UNPINNED with respect to the reference itself.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy_ref as R  # noqa: E402
from mtf_amd import synth  # noqa: E402

SEED = 424242
IMG_SEED = 99
IMG_SHAPE = (192, 192)


def main():
    rng = np.random.default_rng(SEED)
    img = synth.make_frame(*IMG_SHAPE, seed=IMG_SEED)
    out = {"img_seed": IMG_SEED, "img_shape": np.array(IMG_SHAPE)}

    # --- raw sampling / gradient at assorted points, incl. border and integer coordinates
    xs = np.array([-2.0, -1e-9, 0.0, 0.25, 10.0, 10.0 + 1e-8, 190.0, 191.0, 191.5, 195.0, 37.25, 64.0, 100.5, 17.125])
    ys = np.array([5.0, 5.0, 0.0, 0.0, 20.0, 20.0, 30.0, 30.0, 12.0, 7.0, 191.0, 64.0, 33.75, 190.999])
    out["pts"] = np.stack([xs, ys])
    out["pix_vals"] = R.bilinear(img, xs, ys)
    out["img_grad"] = R.img_grad(img, out["pts"])

    # --- homography + SSD, chained warp: config-1 shape (reduced) and a non-rectangular region
    for tag, res, corners in (
        ("sq", 24, synth.square_corners(96, 96, 72)),
        ("quad", 20, synth.square_corners(96, 90, 60) + rng.uniform(-3, 3, size=(2, 4))),
    ):
        init_pts, init_hm = R.grid_from_corners(corners, res, res)
        I0 = R.bilinear(img, init_pts[0], init_pts[1])
        W0 = np.eye(3)
        g0 = R.img_grad(img, init_pts)
        J0 = R.sd_rows_chained(g0, R.hom_spatial_jacobian(W0, init_pts, init_hm[2]),
                               R.hom_param_jacobian(init_pts[0], init_pts[1]))
        p = synth.random_small_homography(rng, 0.5)
        W = R.hom_matrix(p)
        fc = R.lk_step_hom_ssd(img, init_pts, init_hm, W, I0, mode="fclk")
        es = R.lk_step_hom_ssd(img, init_pts, init_hm, W, I0, J0=J0, mode="esm")
        out.update({
            tag + "_res": res, tag + "_corners": corners, tag + "_p": p,
            tag + "_init_pts_head": init_pts[:, :16], tag + "_I0_head": I0[:16], tag + "_J0_head": J0[:16],
            tag + "_It_head": fc["It"][:16], tag + "_grad_head": fc["grad"][:16], tag + "_Jt_head": fc["Jt"][:16],
            tag + "_f": fc["f"], tag + "_fclk_g": fc["g"], tag + "_fclk_H": fc["H"], tag + "_fclk_dp": fc["dp"],
            tag + "_esm_g": es["g"], tag + "_esm_H": es["H"], tag + "_esm_dp": es["dp"],
            # the whole per-pixel arrays of these two cases, not only their first 16 entries (576 and 400 sample points)
            tag + "_init_pts_full": init_pts, tag + "_I0_full": I0, tag + "_J0_full": J0,
            tag + "_It_full": fc["It"], tag + "_grad_full": fc["grad"], tag + "_Jt_full": fc["Jt"],
        })

    # --- affine + NCC (config-3 patch shape)
    res = 25
    corners = synth.square_corners(100, 80, 25)
    init_pts, init_hm = R.grid_from_corners(corners, res, res, affine=True)
    I0 = R.bilinear(img, init_pts[0], init_pts[1])
    g0 = R.img_grad(img, init_pts)
    J0 = R.sd_rows_direct(g0, R.aff_param_jacobian(init_pts[0], init_pts[1]))  # warp = identity
    pa = rng.uniform(-1, 1, 6) * [1.5, 1.5, 0.02, 0.02, 0.02, 0.02]
    A = R.aff_matrix(pa)
    wpts = (A @ np.vstack([init_pts, np.ones(init_pts.shape[1])]))[:2]
    It = R.bilinear(img, wpts[0], wpts[1])
    m = R.ncc(I0, It)
    out.update({"ncc_corners": corners, "ncc_p": pa, "ncc_f": m["f"], "ncc_df_dIt_head": m["df_dIt"][:16],
                "ncc_df_dI0_head": m["df_dI0"][:16], "ncc_g_init": m["df_dI0"] @ J0,
                "ncc_H_self_J0": R.ncc_self_hessian(J0, m), "ncc_J0_head": J0[:16]})

    # --- homography + MI, 8 bins, 40 x 40 (config 5 shape, reduced): similarity, df_dIt . Jt, cmptCurrHessian(Jt)
    res, nb = 40, 8
    corners = synth.square_corners(96, 100, 80) + rng.uniform(-1, 1, size=(2, 4))
    init_pts, init_hm = R.grid_from_corners(corners, res, res)
    mult = (nb - 1) / 256.0                                        # MI.cc:80-94: pixel values scaled to [0, n_bins - 1]
    I0n = mult * R.bilinear(img, init_pts[0], init_pts[1])
    p = synth.random_small_homography(rng, 0.4)
    W = R.hom_matrix(p)
    wpts, q = R.warp_pts(W, init_hm)
    Itn = mult * R.bilinear(img, wpts[0], wpts[1])
    gt = R.img_grad(img, wpts, mult=mult)
    Jt = R.sd_rows_chained(gt, R.hom_spatial_jacobian(W, wpts, q[2]), R.hom_param_jacobian(init_pts[0], init_pts[1]))
    mi_f = R.mi_similarity(I0n, Itn, nb)
    dft = R.mi_curr_grad(I0n, Itn, nb)
    mi_g = dft @ Jt
    for s_ in range(8):   # the analytic gradient against a directional central difference of the definition
        h = 1e-4 / np.abs(Jt[:, s_]).max()
        fd = (R.mi_similarity(I0n, Itn + h * Jt[:, s_], nb) - R.mi_similarity(I0n, Itn - h * Jt[:, s_], nb)) / (2 * h)
        assert abs(fd - mi_g[s_]) <= 1e-6 * max(abs(mi_g[s_]), np.abs(mi_g).max() * 1e-3), (s_, fd, mi_g[s_])
    out.update({"mi_corners": corners, "mi_p": p, "mi_f": mi_f, "mi_g_curr": mi_g, "mi_H_curr": R.mi_curr_hessian(I0n, Itn, Jt, nb),
                "mi_It_head": Itn[:16], "mi_df_dIt_head": dft[:16]})

    # --- PF scores for 32 candidates (config 4 shape, reduced)
    res = 20
    corners = synth.square_corners(90, 100, 60)
    init_pts, init_hm = R.grid_from_corners(corners, res, res)
    I0 = R.bilinear(img, init_pts[0], init_pts[1])
    states = synth.pf_candidate_states(rng, 32)
    lik = []
    for s in states:
        wp, _ = R.warp_pts(R.hom_matrix(s), init_hm)
        r = R.bilinear(img, wp[0], wp[1]) - I0
        lik.append(np.exp(-1.0 * np.sqrt(0.5 * float(r @ r) / I0.size)))
    out.update({"pf_corners": corners, "pf_states": states, "pf_likelihood": np.array(lik)})

    # --- second order: affine + SSD, chained warp, 22 x 22 (FCLK / ESM with sec_ord_hess): the image Hessian is the reference's
    # nine-sample stencil at hess_eps = 1 (imgUtils.cc:334-366) at the warped points; the pixel Hessian of I(A u + t) with
    # respect to the compositional affine parameters is P^T (A2^T Hess A2) P -- the warp is linear in its parameters, so
    # there is no second term (Affine.cc:264-291) -- and SSD's second-order current Hessian adds sum_p df_dIt[p] times it
    # to -Jt^T Jt (SSDBase.cc:334-342, df_dIt = -(It - I0))
    res = 22
    corners = synth.square_corners(98, 92, 66)
    init_pts, init_hm = R.grid_from_corners(corners, res, res, affine=True)
    x, y = init_pts
    I0 = R.bilinear(img, x, y)
    pa = rng.uniform(-1, 1, 6) * [1.2, 1.2, 0.02, 0.02, 0.02, 0.02]
    A = R.aff_matrix(pa)
    wx, wy = (A @ np.vstack([x, y, np.ones_like(x)]))[:2]
    It = R.bilinear(img, wx, wy)
    P = R.aff_param_jacobian(x, y)
    Jt = R.sd_rows_chained(R.img_grad(img, np.stack([wx, wy])), np.broadcast_to(A[:2, :2], (x.size, 2, 2)), P)
    c = R.bilinear(img, wx, wy)
    Hi = np.empty((x.size, 2, 2))
    Hi[:, 0, 0] = (R.bilinear(img, wx + 2, wy) + R.bilinear(img, wx - 2, wy) - 2 * c) / 4
    Hi[:, 1, 1] = (R.bilinear(img, wx, wy + 2) + R.bilinear(img, wx, wy - 2) - 2 * c) / 4
    Hi[:, 0, 1] = Hi[:, 1, 0] = ((R.bilinear(img, wx + 1, wy + 1) + R.bilinear(img, wx - 1, wy - 1)) -
                                 (R.bilinear(img, wx + 1, wy - 1) + R.bilinear(img, wx - 1, wy + 1))) / 4
    Hw = np.einsum("ia,nij,jb->nab", A[:2, :2], Hi, A[:2, :2])
    D = np.einsum("nis,nij,njt->nst", P, Hw, P)
    so_H = -(Jt.T @ Jt) + np.einsum("n,nst->st", -(It - I0), D)
    out.update({"so_corners": corners, "so_p": pa, "so_img_hess_head": Hi[:16].reshape(16, 4), "so_pix_hess_head": D[:8],
                "so_H_curr2": so_H})

    # --- sampler sigmas from a pixel sigma (estimateStateSigma), at a non-identity state, both SSMs
    ess_corners = synth.square_corners(96, 90, 64) + rng.uniform(-2, 2, size=(2, 4))
    ess_p = synth.random_small_homography(rng, 0.4)
    ip, ihm = R.grid_from_corners(ess_corners, 14, 11)
    cp, chm = R.warp_pts(R.hom_matrix(ess_p), ihm)
    ess_pa = rng.uniform(-1, 1, 6) * [1.5, 1.5, 0.03, 0.03, 0.03, 0.03]
    ipa, _ = R.grid_from_corners(ess_corners, 14, 11, affine=True)
    out.update({"ess_corners": ess_corners, "ess_p": ess_p, "ess_pa": ess_pa,
                "ess_hom": R.estimate_state_sigma(ip, cp, chm[2], 1.3), "ess_aff": R.estimate_state_sigma(ipa, None, None, 1.3, affine=True)})

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lk_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
