"""Independent NumPy float64 re-derivation of the LK hot path (TEST INFRASTRUCTURE ONLY).

Written from the mathematics, not from oracle/mtf_oracle.cpp: vectorised bilinear
sampling, central-difference image gradients, chain-rule steepest-descent rows,
Gauss-Newton products through numpy matmul and numpy.linalg.solve.  It exists to
cross-check the C++ restatement and to generate the small golden fixtures under
tests/golden/ (tests/golden/make_golden.py).  The reference itself cannot run here
(Eigen/OpenCV/Boost absent), so these fixtures pin the oracle, not the reference:
PARITY UNPINNED.

Array conventions here are NumPy-natural: pts (2, N), grad (N, 2), J (N, S), H (S, S).
"""
import numpy as np


# ------------------------------------------------------------------ sampling
def bilinear(img, x, y, overflow=128.0):
    """Bilinear interpolant with the constant-border rule: out-of-range points, or points
    whose upper neighbour (taken only when the fractional part is non-zero) leaves the
    image, read `overflow`."""
    h, w = img.shape
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    out = np.full(x.shape, overflow, dtype=np.float64)
    ok = (x >= 0) & (x < w) & (y >= 0) & (y < h)
    xs, ys = np.where(ok, x, 0.0), np.where(ok, y, 0.0)
    lx, ly = np.trunc(xs).astype(np.int64), np.trunc(ys).astype(np.int64)
    dx, dy = xs - lx, ys - ly
    ux = np.where(dx == 0, lx, lx + 1)
    uy = np.where(dy == 0, ly, ly + 1)
    ok &= (ux < w) & (uy < h)
    ux, uy = np.minimum(ux, w - 1), np.minimum(uy, h - 1)
    im = img.astype(np.float64)
    v = ((im[ly, lx] * (1 - dx)) * (1 - dy) + (im[ly, ux] * dx) * (1 - dy) +
         (im[uy, lx] * (1 - dx)) * dy + (im[uy, ux] * dx) * dy)
    return np.where(ok, v, out)


def img_grad(img, pts, eps=1e-8, mult=1.0):
    """Central difference of the interpolant at the given image points: (N, 2)."""
    x, y = pts
    gx = (bilinear(img, x + eps, y) - bilinear(img, x - eps, y)) * (mult / (2 * eps))
    gy = (bilinear(img, x, y + eps) - bilinear(img, x, y - eps)) * (mult / (2 * eps))
    return np.stack([gx, gy], axis=1)


# ------------------------------------------------------------------ warps
def hom_matrix(p):
    return np.array([[1 + p[0], p[1], p[2]], [p[3], 1 + p[4], p[5]], [p[6], p[7], 1.0]])


def aff_matrix(p):
    return np.array([[1 + p[2], p[3], p[0]], [p[4], 1 + p[5], p[1]], [0.0, 0.0, 1.0]])


def dlt(src, dst):
    """Homography taking the 4 src corners (2, 4) to dst (2, 4), h22 = 1."""
    A, b = [], []
    for (x, y), (u, v) in zip(src.T, dst.T):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.array(A, dtype=np.float64), np.array(b, dtype=np.float64))
    return np.append(h, 1.0).reshape(3, 3)


def unit_grid(resx, resy, lo_x=-0.5, lo_y=-0.5, hi_x=0.5, hi_y=0.5):
    xs = np.linspace(lo_x, hi_x, resx)
    ys = np.linspace(lo_y, hi_y, resy)
    gx, gy = np.meshgrid(xs, ys)  # row-major: y outer, x inner
    pts = np.stack([gx.ravel(), gy.ravel()])
    corners = np.array([[lo_x, hi_x, hi_x, lo_x], [lo_y, lo_y, hi_y, hi_y]])
    return pts, corners


def grid_from_corners(corners, resx, resy, affine=False):
    """(2, N) sample grid spanned by the corners, plus its un-normalised homogeneous form."""
    if affine:
        pts, base = unit_grid(resx, resy, 1 - resx / 2.0, 1 - resy / 2.0, resx / 2.0, resy / 2.0)
    else:
        pts, base = unit_grid(resx, resy)
    W = dlt(base, corners)
    hm = W @ np.vstack([pts, np.ones(pts.shape[1])])
    return hm[:2] / hm[2], hm


def warp_pts(W, pts_hm):
    q = W @ pts_hm
    return q[:2] / q[2], q


# ------------------------------------------------------------------ steepest-descent rows
def hom_param_jacobian(x, y):
    """d(Delta W)(x) / d(Delta p) at Delta p = 0 for the 8-dof homography: (N, 2, 8)."""
    z, o = np.zeros_like(x), np.ones_like(x)
    r0 = np.stack([x, y, o, z, z, z, -x * x, -x * y], axis=1)
    r1 = np.stack([z, z, z, x, y, o, -x * y, -y * y], axis=1)
    return np.stack([r0, r1], axis=1)


def aff_param_jacobian(x, y):
    z, o = np.zeros_like(x), np.ones_like(x)
    r0 = np.stack([o, z, x, y, z, z], axis=1)
    r1 = np.stack([z, o, z, z, x, y], axis=1)
    return np.stack([r0, r1], axis=1)


def hom_spatial_jacobian(W, wpts, D):
    """dW(x)/dx of the projective map at each point: (N, 2, 2).
    D is the third homogeneous coordinate as the caller holds it."""
    wx, wy = wpts
    J = np.empty((wx.size, 2, 2))
    J[:, 0, 0] = (W[0, 0] - W[2, 0] * wx) / D
    J[:, 0, 1] = (W[0, 1] - W[2, 1] * wx) / D
    J[:, 1, 0] = (W[1, 0] - W[2, 0] * wy) / D
    J[:, 1, 1] = (W[1, 1] - W[2, 1] * wy) / D
    return J


def sd_rows_chained(grad, spatial_jac, param_jac):
    """dI/dp = grad(I)(w)^T . dW/dx . d(Delta W)/d(Delta p): (N, S)."""
    g = np.einsum("ni,nij->nj", grad, spatial_jac)
    return np.einsum("nj,njs->ns", g, param_jac)


def sd_rows_direct(grad, param_jac):
    return np.einsum("nj,njs->ns", grad, param_jac)


# ------------------------------------------------------------------ similarity measures
def ssd(I0, It):
    r = It - I0
    return dict(f=-0.5 * float(r @ r), df_dI0=r, df_dIt=-r)


def ncc(I0, It):
    a0, at = I0 - I0.mean(), It - It.mean()
    c, b = np.linalg.norm(a0), np.linalg.norm(at)
    f = float(a0 @ at) / (b * c)
    u0, ut = a0 / c, at / b
    dIt = (u0 - f * ut) / b
    dI0 = (ut - f * u0) / c
    return dict(f=f, b=b, c=c, u0=u0, ut=ut, df_dIt=dIt - dIt.mean(), df_dI0=dI0 - dI0.mean())


def ncc_self_hessian(J, m):
    Jc = (J - J.mean(axis=0)) / m["b"]
    v = Jc.T @ m["ut"]
    return -Jc.T @ Jc + np.outer(v, v)


# ------------------------------------------------------------------ one LK step (chained warp)
def lk_step_hom_ssd(img, init_pts, init_hm, W, I0, J0=None, mode="fclk", eps=1e-8):
    """One chained-warp Gauss-Newton step for SSD + homography.

    mode 'fclk': g = -r^T Jt,            H = -Jt^T Jt
    mode 'esm' : g = -r^T (J0 + Jt) / 2, H = -(Jt^T Jt + J0^T J0) / 2   (DiffOfJacs + SumOfSelf)
    Returns dict with It, grad, Jt, f, g, H, dp.
    """
    wpts, q = warp_pts(W, init_hm)
    It = bilinear(img, wpts[0], wpts[1])
    grad = img_grad(img, wpts, eps)
    Jt = sd_rows_chained(grad, hom_spatial_jacobian(W, wpts, q[2]), hom_param_jacobian(init_pts[0], init_pts[1]))
    m = ssd(I0, It)
    if mode == "fclk":
        g = m["df_dIt"] @ Jt
        H = -Jt.T @ Jt
    else:
        g = 0.5 * (m["df_dIt"] @ (J0 + Jt))
        H = 0.5 * (-Jt.T @ Jt - J0.T @ J0)
    dp = -np.linalg.solve(H, g)
    return dict(It=It, grad=grad, Jt=Jt, f=m["f"], g=g, H=H, dp=dp, wpts=wpts)


def compose_hom(W, dp):
    Wn = W @ hom_matrix(dp)
    return Wn / Wn[2, 2]


# ---------------------------------------------------------------------------------------------
# Mutual information with cubic B-spline Parzen windows, from the definition (independent of mtf_oracle.cpp):
#   h_c(r)   = (seed_h + sum_p b3(r - It[p])) * norm          marginal of the current patch
#   h_i(c)   = (seed_h + sum_p b3(c - I0[p])) * norm          marginal of the template
#   h(r, c)  = (seed   + sum_p b3(r - It[p]) b3(c - I0[p])) * norm
#   f        = sum_{r,c} h(r,c) log( h(r,c) / (h_c(r) h_i(c)) )
# with seed = pre_seed, seed_h = n_bins * pre_seed, norm = 1 / (N + seed_h * n_bins)   (AM/src/MI.cc:97-104, 237-262, 369-381).
# The reference's B-spline uses the truncated constant 0.66666666666 (histUtils.h:11); the exact 2/3 here differs by 6.7e-12.
# ---------------------------------------------------------------------------------------------
def bspline3(x):
    ax = np.abs(x)
    out = np.zeros_like(ax)
    m1 = ax < 1
    m2 = (ax >= 1) & (ax < 2)
    out[m1] = 2.0 / 3.0 - ax[m1] ** 2 + ax[m1] ** 3 / 2.0
    out[m2] = (2.0 - ax[m2]) ** 3 / 6.0
    return out


def mi_similarity(I0n, Itn, n_bins=8, pre_seed=10.0):
    """I0n, Itn: patches already scaled to [0, n_bins - 1] (what the AM stores)."""
    I0n = np.asarray(I0n, dtype=np.float64); Itn = np.asarray(Itn, dtype=np.float64)
    N = I0n.size
    bins = np.arange(n_bins, dtype=np.float64)
    Bt = bspline3(bins[:, None] - Itn[None, :])       # n_bins x N
    B0 = bspline3(bins[:, None] - I0n[None, :])
    seed_h = n_bins * pre_seed
    norm = 1.0 / (N + seed_h * n_bins)
    hc = (seed_h + Bt.sum(axis=1)) * norm
    hi = (seed_h + B0.sum(axis=1)) * norm
    hj = (pre_seed + Bt @ B0.T) * norm
    return float(np.sum(hj * np.log(hj / (hc[:, None] * hi[None, :]))))


def bspline3_d1(x):
    ax = np.abs(x); sg = np.sign(x)
    out = np.zeros_like(ax)
    m1 = ax < 1
    m2 = (ax >= 1) & (ax < 2)
    out[m1] = -2.0 * x[m1] + 1.5 * x[m1] * ax[m1]
    out[m2] = -sg[m2] * (2.0 - ax[m2]) ** 2 / 2.0
    return out


def bspline3_d2(x):
    ax = np.abs(x)
    out = np.zeros_like(ax)
    m1 = ax < 1
    m2 = (ax >= 1) & (ax < 2)
    out[m1] = -2.0 + 3.0 * ax[m1]
    out[m2] = 2.0 - ax[m2]
    return out


def mi_curr_hessian(I0n, Itn, J, n_bins=8, pre_seed=10.0):
    """The reference's first-order MI Hessian w.r.t. the current patch (AM/src/MI.cc:603-637), written densely:
    H = J^T diag(t) J + sum_{r,c} (1/h(r,c) - 1/h_c(r)) Q(r,c)^T Q(r,c),
    t_p = sum_r d2/dIt2 b3(r - It_p) norm * sum_c b3(c - I0_p) (1 + log h(r,c) - log h_c(r)),
    Q(r,c) = sum_p d/dIt b3(r - It_p) norm * b3(c - I0_p) * J[p, :].
    (Not the exact second derivative: the 1/h_c term keeps the per-cell outer products -- Dame & Marchand's form.)"""
    I0n = np.asarray(I0n, dtype=np.float64); Itn = np.asarray(Itn, dtype=np.float64)
    N = I0n.size
    bins = np.arange(n_bins, dtype=np.float64)
    X = bins[:, None] - Itn[None, :]
    Bt, B0 = bspline3(X), bspline3(bins[:, None] - I0n[None, :])
    seed_h = n_bins * pre_seed
    norm = 1.0 / (N + seed_h * n_bins)
    hc = (seed_h + Bt.sum(axis=1)) * norm
    hj = (pre_seed + Bt @ B0.T) * norm
    G = 1.0 + np.log(hj) - np.log(hc)[:, None]
    dBt = -bspline3_d1(X) * norm
    d2Bt = bspline3_d2(X) * norm
    t = np.einsum("rp,rc,cp->p", d2Bt, G, B0)
    H = J.T @ (t[:, None] * J)
    Q = np.einsum("rp,cp,ps->rcs", dBt, B0, J)
    fac = 1.0 / hj - 1.0 / hc[:, None]
    return H + np.einsum("rc,rcs,rcu->su", fac, Q, Q)


def mi_curr_grad(I0n, Itn, n_bins=8, pre_seed=10.0):
    """df / dIt per pixel (AM/src/MI.cc:426-442 written densely): sum_{r,c} d/dIt b3(r - It_p) norm * b3(c - I0_p) *
    (1 + log h(r,c) - log h_c(r)); the derivative of b3(r - It) with respect to It is -b3'(r - It)."""
    I0n = np.asarray(I0n, dtype=np.float64); Itn = np.asarray(Itn, dtype=np.float64)
    N = I0n.size
    bins = np.arange(n_bins, dtype=np.float64)
    X = bins[:, None] - Itn[None, :]
    Bt, B0 = bspline3(X), bspline3(bins[:, None] - I0n[None, :])
    seed_h = n_bins * pre_seed
    norm = 1.0 / (N + seed_h * n_bins)
    hc = (seed_h + Bt.sum(axis=1)) * norm
    hj = (pre_seed + Bt @ B0.T) * norm
    G = 1.0 + np.log(hj) - np.log(hc)[:, None]
    return np.einsum("rp,rc,cp->p", -bspline3_d1(X) * norm, G, B0)


# ------------------------------------------------------------------ sampler sigmas from a pixel sigma
def estimate_state_sigma(init_pts, curr_pts, curr_D, pix_sigma, affine=False):
    """StateSpaceModel::estimateStateSigma (SSM/src/ProjectiveBase.cc:201-213): state_sigma[k] = pix_sigma / mean_p |d pt_p / d state_k|, the
    columns being getCurrPixGrad's -- Homography.cc:143-155 (rows [x y 1 0 0 0 -x cx -y cx] / D and [0 0 0 x y 1 -x cy -y cy] / D with
    (x, y) the init point, (cx, cy) the current one, D the current homogeneous denominator), Affine.cc:152-158 ([1 0 x y 0 0], [0 1 0 0 x y])."""
    x, y = init_pts
    if affine:
        r0 = np.stack([np.ones_like(x), np.zeros_like(x), x, y, np.zeros_like(x), np.zeros_like(x)])
        r1 = np.stack([np.zeros_like(x), np.ones_like(x), np.zeros_like(x), np.zeros_like(x), x, y])
    else:
        cx, cy = curr_pts
        z = np.zeros_like(x); o = np.ones_like(x)
        r0 = np.stack([x, y, o, z, z, z, -x * cx, -y * cx]) / curr_D
        r1 = np.stack([z, z, z, x, y, o, -x * cy, -y * cy]) / curr_D
    return pix_sigma / np.sqrt(r0 ** 2 + r1 ** 2).mean(axis=1)


# ------------------------------------------------------------------ r04: second-order Hessians of NCC and MI, PF resampling / estimates, mc::
def ncc_curr_hessian_ref_form(J, m):
    """NCC::cmptCurrHessian (AM/src/NCC.cc:307-335), the reference's FORM: with Jc = (J - column means) / b,
    H = -f Jc^T Jc - (Jc^T ut)(u0^T Jc) - (Jc^T u0)(ut^T Jc) + 3 (Jc^T ut)(ut^T Jc).
    (The exact second derivative of f = u0 . ut has 3 f in the last term -- d2f/dIt2 = [-f P - ut u0^T - u0 ut^T + 3 f ut ut^T] / b^2 --
    the reference writes 3; the two agree at f = 1.  The fixture follows the reference.)"""
    Jc = (J - J.mean(axis=0)) / m["b"]
    vt, v0 = Jc.T @ m["ut"], Jc.T @ m["u0"]
    return -m["f"] * (Jc.T @ Jc) - np.outer(vt, v0) - np.outer(v0, vt) + 3.0 * np.outer(vt, vt)


def ncc_init_hessian_ref_form(J, m):
    """NCC::cmptInitHessian (NCC.cc:282-305): the same four terms with the roles of the two patches swapped in the last one, and --
    a quirk kept on purpose -- the template Jacobian divided by b, the CURRENT patch's norm, not c."""
    Jc = (J - J.mean(axis=0)) / m["b"]
    vt, v0 = Jc.T @ m["ut"], Jc.T @ m["u0"]
    return -m["f"] * (Jc.T @ Jc) - np.outer(vt, v0) - np.outer(v0, vt) + 3.0 * np.outer(v0, v0)


def image_hessian_stencil(img, wx, wy):
    """utils::getImgHess at hess_eps = 1 (Utilities/src/imgUtils.cc:334-366): Ixx, Iyy from samples two pixels apart, Ixy from the four
    diagonal neighbours; (N, 2, 2)."""
    c = bilinear(img, wx, wy)
    Hi = np.empty((np.size(wx), 2, 2))
    Hi[:, 0, 0] = (bilinear(img, wx + 2, wy) + bilinear(img, wx - 2, wy) - 2 * c) / 4
    Hi[:, 1, 1] = (bilinear(img, wx, wy + 2) + bilinear(img, wx, wy - 2) - 2 * c) / 4
    Hi[:, 0, 1] = Hi[:, 1, 0] = ((bilinear(img, wx + 1, wy + 1) + bilinear(img, wx - 1, wy - 1)) -
                                 (bilinear(img, wx + 1, wy - 1) + bilinear(img, wx - 1, wy + 1))) / 4
    return Hi


def affine_pix_hessian(Hi, A2, P):
    """d2 I(A u + t) / dp2 for the compositional affine parameters: P^T (A2^T Hess A2) P per pixel (the warp is linear in its
    parameters: no second term, SSM/src/Affine.cc:264-291); (N, S, S)."""
    Hw = np.einsum("ia,nij,jb->nab", A2, Hi, A2)
    return np.einsum("nis,nij,njt->nst", P, Hw, P)


def _mi_tables(I0n, Itn, n_bins, pre_seed):
    N = I0n.size
    bins = np.arange(n_bins, dtype=np.float64)
    X = bins[:, None] - Itn[None, :]
    Bt, B0 = bspline3(X), bspline3(bins[:, None] - I0n[None, :])
    seed_h = n_bins * pre_seed
    norm = 1.0 / (N + seed_h * n_bins)
    hc = (seed_h + Bt.sum(axis=1)) * norm
    return X, Bt, B0, norm, hc


def mi_self_hessian2(Itn, J, D, n_bins=8, pre_seed=10.0):
    """MI::cmptSelfHessian with pixel Hessians (AM/src/MI.cc:697-735), densely: the joint histogram of the current patch WITH ITSELF
    (cmptSelfHist :639-657), G = 1 + log h_self(r, t) - log h_c(r);
    H = sum_p [ (sum_r b3''(r - It_p) norm sum_t b3(t - It_p) G(r, t)) J_p^T J_p + (sum_r d/dIt b3(r - It_p) norm sum_t b3(t - It_p) G(r, t)) D_p ]
        + sum_{r,t} (1 / h_self(r, t) - 1 / h_c(r)) Q(r, t)^T Q(r, t),   Q(r, t) = sum_p d/dIt b3(r - It_p) norm b3(t - It_p) J_p."""
    Itn = np.asarray(Itn, dtype=np.float64)
    X, Bt, _, norm, hc = _mi_tables(Itn, Itn, n_bins, pre_seed)
    hs = (pre_seed + Bt @ Bt.T) * norm
    G = 1.0 + np.log(hs) - np.log(hc)[:, None]
    dBt = -bspline3_d1(X) * norm
    d2Bt = bspline3_d2(X) * norm
    inner = np.einsum("rt,tp->rp", G, Bt)
    hess_term = np.einsum("rp,rp->p", d2Bt, inner)
    grad_term = np.einsum("rp,rp->p", dBt, inner)
    H = J.T @ (hess_term[:, None] * J) + np.einsum("p,pst->st", grad_term, D)
    Q = np.einsum("rp,tp,ps->rts", dBt, Bt, J)
    fac = 1.0 / hs - 1.0 / hc[:, None]
    return H + np.einsum("rt,rts,rtu->su", fac, Q, Q)


# --- particle filter: weights -> cumulative weights -> resampling -> estimate (SM/src/NT/PF.cc:345-614)
def pf_multinomial_ids(w, u):
    """binary / linear multinomial resampling (NT/PF.cc:455-536): particle k takes the smallest index whose NORMALISED cumulative weight
    reaches its draw u_k"""
    cum = np.cumsum(w)
    return np.searchsorted(cum / cum[-1], u, side="left").clip(0, len(w) - 1)


def pf_residual_ids(w):
    """residual resampling (NT/PF.cc:538-582): weights normalised; the indices EXCEPT THE LAST ONE sorted by weight, highest first
    (std::sort(idx, idx + n - 1): the range ends one short; ties: index order is one of its outcomes -- the fixture has no ties);
    every particle in that order copied round(w n) times until n slots are filled, the rest take the first of the order"""
    n = len(w)
    wn = w / w.sum()
    order = list(np.argsort(-wn[:n - 1], kind="stable")) + [n - 1]
    ids = []
    for i in order:
        c = int(np.floor(wn[i] * n + 0.5))   # C round(): half away from zero; the weights are positive
        ids.extend([i] * c)
        if len(ids) >= n:
            break
    ids = ids[:n] + [order[0]] * max(0, n - len(ids))
    return np.array(ids), order[0]


def running_mean(rows):
    """mean += (x - mean) / (k + 1) (ProjectiveBase::estimateMeanOfSamples ProjectiveBase.cc:313-319, PF::updateMeanCorners NT/PF.cc:607-614)"""
    m = np.zeros_like(np.asarray(rows[0], dtype=np.float64))
    for k, x in enumerate(rows):
        m = m + (x - m) / (k + 1)
    return m


# --- mc:: sampling (Utilities/src/imgUtils.cc:861-1005): the channels of a 32FC3 frame are sampled independently, rows = (pixel, channel)
def mc_pix_vals(img3, pts):
    """(N * C,) in (pixel, channel) order"""
    C = img3.shape[2]
    return np.stack([bilinear(img3[:, :, c], pts[0], pts[1]) for c in range(C)], axis=1).ravel()


def mc_img_grad(img3, pts, eps=1e-8):
    """(N * C, 2) in (pixel, channel) order"""
    C = img3.shape[2]
    g = np.stack([img_grad(img3[:, :, c], pts, eps) for c in range(C)], axis=1)   # (N, C, 2)
    return g.reshape(-1, 2)


# ---------------------------------------------------------------------------------------------
# r05 fixtures (tests/golden/make_golden3.py): GridTracker's patch layout, NN dataset rows, the stochastic samplers with given draws
# ---------------------------------------------------------------------------------------------
def grid_layout(region, grid_x, grid_y, patch_x, patch_y, dyn_patch_size, patch_centroid_inside):
    """GridTracker::resetTrackers' geometry (SM/src/GridTracker.cc:345-380) from the description of the algorithm: a grid SSM with
    (grid + 1)^2 points when patches are built from the cells' corners (dyn_patch_size or patch_centroid_inside), else grid^2 points;
    its points are the uniform grid of the unit square pushed through the homography that takes the square's corners to the region's;
    patch (r, c) is the cell's quadrilateral, or the patch-size rectangle centred on the cell's centroid / on grid point (r, c).
    -> (grid points (n_pts, 2), patch corners (n, 2, 4) TL TR BR BL)."""
    extra = 1 if (dyn_patch_size or patch_centroid_inside) else 0
    rx, ry = grid_x + extra, grid_y + extra
    pts, _ = grid_from_corners(np.asarray(region, dtype=np.float64), rx, ry)
    P = pts.T.reshape(ry, rx, 2)
    out = np.empty((grid_x * grid_y, 2, 4))
    for r in range(grid_y):
        for c in range(grid_x):
            k = r * grid_x + c
            if extra:
                quad = np.stack([P[r, c], P[r, c + 1], P[r + 1, c + 1], P[r + 1, c]], axis=1)   # (2, 4)
            if dyn_patch_size:
                out[k] = quad
                continue
            ctr = quad.mean(axis=1) if patch_centroid_inside else P[r, c]
            x0, y0 = ctr[0] - patch_x / 2.0, ctr[1] - patch_y / 2.0
            out[k] = [[x0, x0 + patch_x, x0 + patch_x, x0], [y0, y0, y0 + patch_y, y0 + patch_y]]
    return pts.T.copy(), out


def nn_dataset_rows(img, init_hm, perts, ncc=False):
    """NN::generateDataset (SM/src/NT/NN.cc:131-191) for a homography SSM with compositional updates: per sample the SSM is moved by
    the INVERSE of the perturbation (invertState: inverse matrix scaled to h22 = 1; compositionalUpdate: right-multiply, rescale),
    the patch is sampled there and turned into the AM's distance feature -- SSD: the pixel values (SSDBase.h:116-125); NCC: centred
    and scaled to unit norm (NCC.cc:530-537) -- and the SSM is moved back by the perturbation itself, so that the warp the next sample
    starts from is the accumulated product, not exactly the identity.  init_hm: (3, N) homogeneous template grid."""
    W = np.eye(3)
    rows = []
    for p in perts:
        Wp = hom_matrix(p)
        inv = np.linalg.inv(Wp)
        inv = inv / inv[2, 2]
        W = W @ inv
        W = W / W[2, 2]
        q = W @ init_hm
        It = bilinear(img, q[0] / q[2], q[1] / q[2])
        if ncc:
            It = It - It.mean()
            It = It / np.linalg.norm(It)
        rows.append(It)
        W = W @ Wp
        W = W / W[2, 2]
    return np.stack(rows)


def nn_mi_dist_feat(It, n_bins, pou):
    """MI::updateDistFeat (AM/src/MI.cc:736-747) of raw pixel values It (N,): the AM's pixel normalisation first (MI.cc:80-94: [0, 255] ->
    [0, n_bins - 1], with partition of unity [1, n_bins - 2], over PIX_MAX - PIX_MIN + 1 = 256), then per pixel the 5 x N matrix, row-major:
    floor(v) | bSpl3(d), bSpl3(d + 1), bSpl3(d + 2), bSpl3(d + 3) with d = first id of the floor's B-spline window (max(floor - 1, 0), the
    standard ids of histUtils) - v.  -> (5 N,)"""
    lo, hi = (1.0, n_bins - 2.0) if pou else (0.0, n_bins - 1.0)
    v = (hi - lo) / 256.0 * np.asarray(It, dtype=np.float64) + lo
    fl = np.floor(v)
    d = np.maximum(fl - 1.0, 0.0) - v
    return np.concatenate([fl, bspline3(d), bspline3(d + 1.0), bspline3(d + 2.0), bspline3(d + 3.0)])


def hom_corner_sampler(init_corners, sigma, mean, z):
    """Homography::generatePerturbation with corner based sampling (SSM/src/Homography.cc:899-909): one translation from distribution 0
    (two draws), eight corner offsets from distribution 1, the state of the 4-point homography init_corners -> disturbed corners.
    z: ten standard normals in the order the reference draws them (t_x, t_y, then x, y per corner)."""
    t = mean[0] + sigma[0] * z[:2]
    d = (mean[1] + sigma[1] * z[2:10]).reshape(4, 2).T
    H = dlt(init_corners, init_corners + d + t[:, None])
    return np.array([H[0, 0] - 1, H[0, 1], H[0, 2], H[1, 0], H[1, 1] - 1, H[1, 2], H[2, 0], H[2, 1]])


def aff_point_sampler(init_corners, mode, sigma, mean, z):
    """Affine::generatePerturbation with pt_based_sampling 1 / 2 (SSM/src/Affine.cc:464-494): the bottom-right corner, the bottom-left
    corner and the centre of the top edge are disturbed -- mode 1: coordinate j by distribution j (six draws); mode 2: every coordinate
    by distribution 1 (six draws), then one translation by distribution 0 (two draws) -- and the perturbation is the affine map of the
    three point pairs, as a state [tx, ty, a - 1, b, c, d - 1]."""
    c = np.asarray(init_corners, dtype=np.float64)
    orig = np.stack([c[:, 2], c[:, 3], (c[:, 0] + c[:, 1]) / 2.0], axis=1)     # (2, 3)
    if mode == 1:
        pert = orig + (np.asarray(mean[:6]) + np.asarray(sigma[:6]) * z[:6]).reshape(3, 2).T
    else:
        pert = orig + (mean[1] + sigma[1] * z[:6]).reshape(3, 2).T + (mean[0] + sigma[0] * z[6:8])[:, None]
    A = np.vstack([orig, np.ones(3)])                                           # 3 x 3, exact for three non-collinear points
    M = pert @ np.linalg.inv(A)                                                 # 2 x 3
    return np.array([M[0, 2], M[1, 2], M[0, 0] - 1, M[0, 1], M[1, 0], M[1, 1] - 1])


# ---------------------------------------------------------------------------------------------
# r06 fixtures (tests/golden/make_golden4.py): GridTracker's forward-backward mask
# ---------------------------------------------------------------------------------------------
def grid_fb_mask(prev_pts, curr_pts, fb_prev_pts, fb_err_thresh, n_model_pts):
    """The selection half of GridTracker::backwardEstimation (SM/src/GridTracker.cc:307-332) from its description: a patch is kept when the
    squared distance between the centroid its backward track reached and the one it started from (single-precision points: the
    coordinate differences are single-precision numbers, their squares and the sum are not) does not exceed the threshold; when fewer
    than n_model_pts are kept, rejected patches are re-admitted in index order until that many are there, and their pairs go BEHIND
    the kept ones.  -> (mask (n,) bool, prev pairs (c, 2) float32, curr pairs (c, 2) float32)."""
    a = np.asarray(prev_pts, dtype=np.float32)
    b = np.asarray(curr_pts, dtype=np.float32)
    fb = np.asarray(fb_prev_pts, dtype=np.float32)
    d = (fb - a).astype(np.float64)                      # float32 subtraction, widened afterwards
    keep = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] <= fb_err_thresh
    order = list(np.flatnonzero(keep))
    if len(order) < n_model_pts:
        extra = list(np.flatnonzero(~keep)[:n_model_pts - len(order)])
        order += extra
        keep = keep.copy()
        keep[extra] = True
    idx = np.asarray(order, dtype=np.int64)
    return keep, a[idx], b[idx]
