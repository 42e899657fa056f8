#!/usr/bin/env python
"""Generates tests/golden/lk_golden2.npz (r04): the paths where the C++ oracle and the HIP product could still share a mistake,
from the independent NumPy re-derivation (oracle/numpy_ref.py) -- NCC and MI second-order Hessians (AM/src/NCC.cc:391-410,
AM/src/MI.cc:659-735), particle-filter resampling and estimates (SM/src/NT/PF.cc:345-614), one multi-channel sampling / gradient case
(Utilities/src/imgUtils.cc:861-1005).  A separate file so that lk_golden.npz stays byte-identical to its generator.
PARITY UNPINNED with respect to the reference itself (it ships no vectors and cannot be built here).

Run from the repo root:  python tests/golden/make_golden2.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy_ref as R  # noqa: E402
from mtf_amd import synth  # noqa: E402

SEED = 20260929
IMG_SEED = 99
IMG_SHAPE = (192, 192)


def main():
    rng = np.random.default_rng(SEED)
    img = synth.make_frame(*IMG_SHAPE, seed=IMG_SEED)
    out = {"img_seed": IMG_SEED, "img_shape": np.array(IMG_SHAPE)}

    # ---- affine, chained warp, 22 x 22: the pieces both second-order cases share
    res = 22
    corners = synth.square_corners(97, 93, 66)
    init_pts, _ = R.grid_from_corners(corners, res, res, affine=True)
    x, y = init_pts
    P = R.aff_param_jacobian(x, y)
    pa = rng.uniform(-1, 1, 6) * [1.1, 1.1, 0.02, 0.02, 0.02, 0.02]
    A = R.aff_matrix(pa)
    wx, wy = (A @ np.vstack([x, y, np.ones_like(x)]))[:2]
    A2 = np.broadcast_to(A[:2, :2], (x.size, 2, 2))

    # NCC (pixel values as sampled)
    I0 = R.bilinear(img, x, y); It = R.bilinear(img, wx, wy)
    J0 = R.sd_rows_direct(R.img_grad(img, init_pts), P)
    Jt = R.sd_rows_chained(R.img_grad(img, np.stack([wx, wy])), A2, P)
    D0 = R.affine_pix_hessian(R.image_hessian_stencil(img, x, y), np.eye(2), P)
    Dt = R.affine_pix_hessian(R.image_hessian_stencil(img, wx, wy), A[:2, :2], P)
    m = R.ncc(I0, It)
    out.update({"so2_corners": corners, "so2_p": pa,
                "ncc_H_curr1": R.ncc_curr_hessian_ref_form(Jt, m), "ncc_H_init1": R.ncc_init_hessian_ref_form(J0, m),
                # NCC::cmptCurrHessian / cmptInitHessian with pixel Hessians (NCC.cc:391-410): first order + sum_p df_dI[p] D[p]
                "ncc_H_curr2": R.ncc_curr_hessian_ref_form(Jt, m) + np.einsum("n,nst->st", m["df_dIt"], Dt),
                "ncc_H_init2": R.ncc_init_hessian_ref_form(J0, m) + np.einsum("n,nst->st", m["df_dI0"], D0)})

    # MI, 8 bins: pixel values scaled to [0, n_bins - 1] (MI.cc:80-94), gradients and Hessians scale with them
    nb = 8
    mult = (nb - 1) / 256.0
    I0n, Itn = mult * I0, mult * It
    Jtn = mult * Jt
    Dtn = mult * Dt
    dft = R.mi_curr_grad(I0n, Itn, nb)
    out.update({"mi_H_curr2": R.mi_curr_hessian(I0n, Itn, Jtn, nb) + np.einsum("n,nst->st", dft, Dtn),   # MI.cc:679-695
                "mi_H_self2": R.mi_self_hessian2(Itn, Jtn, Dtn, nb),                                      # MI.cc:697-735
                "mi_H_self1": R.mi_self_hessian2(Itn, Jtn, np.zeros_like(Dtn), nb)})                      # MI.cc:515-601 (first order)

    # ---- particle filter: 48 particles on a 20 x 20 homography template, SSD likelihood alpha = 2; proposals = the given states
    # (zero draws), weights -> multinomial / residual resampling -> the three estimates
    n, res = 48, 20
    corners = synth.square_corners(92, 98, 60)
    init_pts, init_hm = R.grid_from_corners(corners, res, res)
    I0 = R.bilinear(img, init_pts[0], init_pts[1])
    states = synth.pf_candidate_states(rng, n) * 0.6
    alpha = 2.0
    w = []
    for s in states:
        wp, _ = R.warp_pts(R.hom_matrix(s), init_hm)
        r = R.bilinear(img, wp[0], wp[1]) - I0
        w.append(np.exp(-alpha * np.sqrt(0.5 * float(r @ r) / I0.size)))   # SSD::getLikelihood SSD.h:41-43
    w = np.array(w)
    u = rng.uniform(0.0, 1.0, n)
    ids = R.pf_multinomial_ids(w, u)
    new = states[ids]
    chm = np.vstack([corners, np.ones(4)])
    warped = [R.warp_pts(R.hom_matrix(s), chm)[0] for s in new]
    rid, rbest = R.pf_residual_ids(w)
    assert len(set(np.round(w / w.sum() * n, 12))) == n   # no ties: std::sort's order is determined
    out.update({"pf2_corners": corners, "pf2_states": states, "pf2_alpha": alpha, "pf2_uniforms": u, "pf2_wts": w,
                "pf2_ids_multinomial": ids, "pf2_max_wt_id_new_set": int(max(range(n), key=lambda k: (w[ids[k]], k))),
                "pf2_mean_state": R.running_mean(list(new)),            # mean_type SSM: ProjectiveBase::estimateMeanOfSamples
                "pf2_mean_corners": R.running_mean(warped),             # mean_type Corners: PF::updateMeanCorners
                "pf2_ids_residual": rid, "pf2_residual_best": int(rbest)})

    # ---- mc:: a 3-channel frame, 12 x 10 points of a warped homography grid: values and central-difference gradients per channel
    img3 = synth.make_frame_mc(96, 96, seed=IMG_SEED + 7)
    corners = synth.square_corners(48, 50, 40) + rng.uniform(-1.5, 1.5, size=(2, 4))
    ip, ihm = R.grid_from_corners(corners, 12, 10)
    p = synth.random_small_homography(rng, 0.3)
    wp, _ = R.warp_pts(R.hom_matrix(p), ihm)
    out.update({"mc_img_seed": IMG_SEED + 7, "mc_corners": corners, "mc_p": p, "mc_I0": R.mc_pix_vals(img3, ip),
                "mc_It": R.mc_pix_vals(img3, wp), "mc_dIt_dx": R.mc_img_grad(img3, wp)})

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lk_golden2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
