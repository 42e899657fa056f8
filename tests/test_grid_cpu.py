"""GridTracker's patch layout (SM/src/GridTracker.cc:86-94 updateRes, :139-146 _linear_idx, :345-380 resetTrackers) -- host arithmetic
of the C ABI (mtfhip_grid_res / mtfhip_grid_layout need no device) against the oracle's restatement, in all three patch modes, for
square, quadrilateral and strongly projective regions and both grid SSMs."""
import ctypes as C

import numpy as np
import pytest

from mtf_amd import _lib as L


def _lib():
    return C.CDLL(L.LIB_PATH)


def _layout(lib, gd, region):
    rx, ry = C.c_int(), C.c_int()
    assert lib.mtfhip_grid_res(C.byref(gd), C.byref(rx), C.byref(ry)) == 0
    n = gd.grid_size_x * gd.grid_size_y
    pts, pcs = np.empty((rx.value * ry.value, 2)), np.empty((n, 4, 2))
    r = np.ascontiguousarray(np.asarray(region, dtype=np.float64).T)
    rc = lib.mtfhip_grid_layout(C.byref(gd), r.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p), pcs.ctypes.data_as(C.c_void_p))
    return rc, (rx.value, ry.value), pts, pcs.transpose(0, 2, 1)


REGIONS = {
    "square": np.array([[100.0, 400, 400, 100], [120, 120, 420, 420]]),
    "parallelogram": np.array([[100.0, 400, 440, 140], [120, 150, 420, 390]]),
    "quad": np.array([[103.0, 398, 405, 96], [118.5, 122.5, 424, 417]]),
    "projective": np.array([[100.0, 420, 360, 150], [120, 90, 400, 330]]),
}


@pytest.mark.parametrize("mode", [(0, 1), (0, 0), (1, 0), (1, 1)], ids=["centroid_inside", "grid_points", "dyn_patch", "dyn+inside"])
@pytest.mark.parametrize("region", sorted(REGIONS))
@pytest.mark.parametrize("grid_ssm", [0, 1], ids=["hom", "aff"])
def test_patch_layout_matches_oracle(oracle, mode, region, grid_ssm):
    dyn, inside = mode
    lib = _lib()
    # (a 1 x 1 grid SSM -- grid_size 1 without the extra row -- is degenerate in the reference as well: Affine's normalised square collapses)
    for (gx, gy, px, py) in ((16, 16, 25, 25), (10, 10, 10, 10), (5, 3, 12, 31), (1, 1, 9, 9) if (dyn or inside) else (2, 2, 9, 9)):
        gd = L.GridDesc(gx, gy, px, py, 1, dyn, inside)
        rc, res, pts, pcs = _layout(lib, gd, REGIONS[region])
        assert rc == 0
        gp = oracle.GridParams(gx, gy, px, py, 1, dyn, inside)
        assert res == oracle.grid_res(gp) == ((gx + 1, gy + 1) if (dyn or inside) else (gx, gy))
        ssm = oracle.SSM(grid_ssm, res[0], res[1])
        g = oracle.Grid(ssm, grid_size=gx, grid_size_y=gy, patch_size=px, patch_size_y=py, dyn_patch_size=dyn, patch_centroid_inside=inside)
        g.initialize(REGIONS[region])
        np.testing.assert_allclose(pts, ssm.get("curr_pts").reshape(-1, 2), rtol=0, atol=1e-9)     # ssm.getPts() of the grid SSM
        np.testing.assert_allclose(pcs, g.patch_corners(), rtol=0, atol=1e-9)
        # utils::getCentroid into cv::Point2f: the centroids resetTrackers leaves in prev_pts are floats
        want = (g.patch_corners().sum(axis=2) / 4.0).astype(np.float32)
        np.testing.assert_allclose(g.prev_pts(), want, rtol=0, atol=4e-5)
        if not dyn:   # axis-aligned patch_size rectangles in cv::Rect corner order
            assert np.allclose(pcs[:, 0, 1] - pcs[:, 0, 0], px) and np.allclose(pcs[:, 1, 2] - pcs[:, 1, 1], py)
            assert np.array_equal(pcs[:, 0, 0], pcs[:, 0, 3]) and np.array_equal(pcs[:, 1, 0], pcs[:, 1, 1])


def test_layout_is_projective_not_bilinear(oracle):
    """for a region that is not a parallelogram the grid SSM's points are NOT the bilinear interpolation of the region's corners
    (what the r04 layout used): they differ by pixels on a strongly projective quadrilateral"""
    gd = L.GridDesc(8, 8, 10, 10, 1, 0, 1)
    _, res, pts, _ = _layout(_lib(), gd, REGIONS["projective"])
    c = REGIONS["projective"]
    u = np.linspace(0.0, 1.0, res[0])
    bil = np.array([[(c[:, 0] + (c[:, 1] - c[:, 0]) * uu) * (1 - vv) + (c[:, 3] + (c[:, 2] - c[:, 3]) * uu) * vv for uu in u] for vv in u]).reshape(-1, 2)
    assert np.abs(bil - pts).max() > 3.0
    _, _, pts_sq, _ = _layout(_lib(), gd, REGIONS["parallelogram"])
    c = REGIONS["parallelogram"]
    bil = np.array([[(c[:, 0] + (c[:, 1] - c[:, 0]) * uu) * (1 - vv) + (c[:, 3] + (c[:, 2] - c[:, 3]) * uu) * vv for uu in u] for vv in u]).reshape(-1, 2)
    np.testing.assert_allclose(pts_sq, bil, atol=1e-9)


def test_layout_refusals():
    lib = _lib()
    lib.mtfhip_last_error.restype = C.c_char_p
    gd = L.GridDesc(4, 4, 10, 10, 1, 0, 1)
    rc, _, _, _ = _layout(lib, gd, np.array([[100.0, 110, 120, 130], [200.0, 200, 200, 200]]))   # collinear corners
    assert rc != 0 and b"degenerate" in lib.mtfhip_last_error()
    bad = L.GridDesc(0, 4, 10, 10, 1, 0, 1)
    rx, ry = C.c_int(), C.c_int()
    assert lib.mtfhip_grid_res(C.byref(bad), C.byref(rx), C.byref(ry)) != 0


def test_oracle_grid_refuses_what_the_reference_constructor_throws_for(oracle):
    ssm = oracle.SSM(0, 5, 5)
    with pytest.raises(ValueError):
        oracle.Grid(ssm, grid_size=5, patch_size=10)           # needs a 6 x 6 SSM with patch_centroid_inside (GridTracker.cc:130-134)
    oracle.Grid(ssm, grid_size=5, patch_size=10, patch_centroid_inside=0)
    oracle.Grid(ssm, grid_size=4, patch_size=10)
