import numpy as np, sys
sys.path.insert(0, '/root/repo')
from mtf_amd import synth
from oracle import numpy_ref as R
img = synth.make_frame(512, 512)
im = img.astype(np.float64)
eps = 1e-8
def bil(x, y):
    lx = np.trunc(x).astype(int); ly = np.trunc(y).astype(int)
    dx = x - lx; dy = y - ly
    return im[ly, lx]*(1-dx)*(1-dy) + im[ly, lx+1]*dx*(1-dy) + im[ly+1, lx]*(1-dx)*dy + im[ly+1, lx+1]*dx*dy
def rel(a, b): return np.linalg.norm(a-b)/np.linalg.norm(b)
rng = np.random.default_rng(3)
for res, size in ((60, 120.0), (200, 200.0)):
    c = synth.square_corners(250.0, 262.0, size)
    pts, hm = R.grid_from_corners(c, res, res)
    p = synth.random_small_homography(rng, 0.6)
    W = synth.homography_from_state(p)
    # warp about region? keep simple: W applied to normalised? use hm from DLT (unnormalised)
    Wd = R.dlt(R.unit_grid(res,res)[1], c)
    Wc = W @ Wd if False else Wd.copy()
    Wc[0,:] += 0.013*Wc[2,:]; Wc[2,0] += 3e-4; Wc[2,1] -= 2e-4; Wc[0,1]+=0.02*size
    g, _ = R.unit_grid(res, res)
    ih = np.vstack([g, np.ones(g.shape[1])])
    q = Wc @ ih
    cx, cy, D = q
    wx, wy = cx / D, cy / D
    lx = np.trunc(wx); ly = np.trunc(wy)
    t00 = im[ly.astype(int), lx.astype(int)]; t01 = im[ly.astype(int), lx.astype(int)+1]
    t10 = im[ly.astype(int)+1, lx.astype(int)]; t11 = im[ly.astype(int)+1, lx.astype(int)+1]
    a = t01 - t00; b = t10 - t00; cc = (t11 - t10) - a
    dx = wx - lx; dy = wy - ly
    # chained
    gx_fd = (bil(wx+eps, wy) - bil(wx-eps, wy)) / (2*eps)
    gy_fd = (bil(wx, wy+eps) - bil(wx, wy-eps)) / (2*eps)
    gx_cf = a + cc*dy; gy_cf = b + cc*dx
    hx = (wx+eps) - (wx-eps); hy = (wy+eps) - (wy-eps)
    gx_q = gx_cf * hx / (2*eps); gy_q = gy_cf * hy / (2*eps)
    def sums(gx, gy): return np.array([np.sum(gx*gx), np.sum(gx*gy), np.sum(gy*gy), np.sum(gx*wx), np.sum(gy)])
    print(res, "chained  closed vs fd", rel(sums(gx_cf, gy_cf), sums(gx_fd, gy_fd)), " quant vs fd", rel(sums(gx_q, gy_q), sums(gx_fd, gy_fd)))
    # unchained homography updateGradPts
    ex0, ex1, ex2 = Wc[0,0]*eps, Wc[1,0]*eps, Wc[2,0]*eps
    ey0, ey1, ey2 = Wc[0,1]*eps, Wc[1,1]*eps, Wc[2,1]*eps
    def fd(e0, e1, e2):
        n0x = cx + e0; n0y = cy + e1; d0 = D + e2
        n1x = cx - e0; n1y = cy - e1; d1 = D - e2
        return (bil(n0x/d0, n0y/d0) - bil(n1x/d1, n1y/d1)) / (2*eps), (n0x, n0y, d0, n1x, n1y, d1)
    ux_fd, tx = fd(ex0, ex1, ex2); uy_fd, ty = fd(ey0, ey1, ey2)
    # closed: directional derivative along dW/dx column
    inv = 1/D
    def cf(e0, e1, e2):   # exact math
        ddx = (e0 - wx*e2)*inv/eps; ddy = (e1 - wy*e2)*inv/eps
        return gx_cf*ddx + gy_cf*ddy
    ux_cf = cf(ex0, ex1, ex2); uy_cf = cf(ey0, ey1, ey2)
    def q(tt):
        n0x, n0y, d0, n1x, n1y, d1 = tt
        hd0 = d0 - D; hd1 = d1 - D
        hnx0 = n0x - cx; hnx1 = n1x - cx; hny0 = n0y - cy; hny1 = n1y - cy
        inv2 = inv*inv
        dpx = (cx*(hd1-hd0) + D*(hnx0-hnx1) + hnx0*hd1 - hnx1*hd0) * inv2
        dpy = (cy*(hd1-hd0) + D*(hny0-hny1) + hny0*hd1 - hny1*hd0) * inv2
        # inc - dec = a*dpx + b*dpy + c*(x0*y0 - x1*y1) (cell-relative); x0*y0-x1*y1 = dpx*ym + dpy*xm with means ~ dx,dy
        return (a*dpx + b*dpy + cc*(dpx*dy + dpy*dx)) / (2*eps)
    ux_q = q(tx); uy_q = q(ty)
    print(res, "unchained closed vs fd", rel(sums(ux_cf, uy_cf), sums(ux_fd, uy_fd)), " quant vs fd", rel(sums(ux_q, uy_q), sums(ux_fd, uy_fd)))
