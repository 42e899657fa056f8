#!/usr/bin/env python
"""Generates tests/golden/lk_golden4.npz (r06) from the independent NumPy re-derivation (oracle/numpy_ref.py): the selection half of
GridTracker's forward-backward error estimation (SM/src/GridTracker.cc:307-332) -- cases where every patch survives, where some are
rejected, where fewer than n_model_pts survive and the set is filled up in tracker order, where none survives, and distances that sit
on the threshold to the last single-precision bit; and nt::NN's dataset rows for the MI appearance model (MI.cc:736-747) from lk_golden3's raw rows.
PARITY UNPINNED with respect to the reference itself (it ships no vectors and cannot be built here).

Run from the repo root:  python tests/golden/make_golden4.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy_ref as R  # noqa: E402

SEED = 20261001


def main():
    rng = np.random.default_rng(SEED)
    out = {}
    cases = []
    n = 36
    prev = rng.uniform(40, 400, size=(n, 2)).astype(np.float32)
    curr = (prev + rng.normal(0, 1.5, size=(n, 2))).astype(np.float32)
    # (name, noise of the round trip in px, threshold, n_model_pts)
    for name, noise, thresh, nm in (("all_kept", 0.05, 2.0, 4), ("some_rejected", 1.0, 2.0, 4), ("few_kept_filled", 3.0, 0.5, 8), ("none_kept_filled", 30.0, 0.01, 4),
                                    ("n_model_pts_3", 2.0, 1.0, 3)):
        fb = (prev + rng.normal(0, noise, size=(n, 2))).astype(np.float32)
        cases.append((name, prev, curr, fb, thresh, nm))
    # on the threshold: offsets whose squared length is exactly 2 in double once the float difference has been taken (1, 1), and one float ulp either side
    p2 = np.tile(np.array([[100.0, 200.0]], dtype=np.float32), (6, 1))
    fb2 = p2.copy()
    fb2[0] += np.array([1.0, 1.0], dtype=np.float32)
    fb2[1] += np.array([1.0, np.nextafter(np.float32(1.0), np.float32(2.0))], dtype=np.float32)
    fb2[2] += np.array([1.0, np.nextafter(np.float32(1.0), np.float32(0.0))], dtype=np.float32)
    fb2[3] += np.array([-1.0, 1.0], dtype=np.float32)
    fb2[4] += np.array([np.sqrt(np.float32(2.0)), 0.0], dtype=np.float32)
    fb2[5] += np.array([0.0, -1.4142135], dtype=np.float32)
    cases.append(("on_threshold", p2, (p2 + 0.25).astype(np.float32), fb2, 2.0, 2))
    out["fb_case_names"] = np.array([c[0] for c in cases])
    for name, prev_c, curr_c, fb_c, thresh, nm in cases:
        mask, pm, cm = R.grid_fb_mask(prev_c, curr_c, fb_c, thresh, nm)
        out.update({"fb_%s_prev" % name: prev_c, "fb_%s_curr" % name: curr_c, "fb_%s_fb" % name: fb_c, "fb_%s_params" % name: np.array([thresh, nm]),
                    "fb_%s_mask" % name: mask, "fb_%s_prev_masked" % name: pm, "fb_%s_curr_masked" % name: cm})
    # ---- NN dataset rows of the MI appearance model (MI.cc:736-747): lk_golden3's raw rows (its perturbations, its template) through the AM's pixel
    # normalisation and updateDistFeat, for the reference's default (8 bins) and the shipped configuration (10 bins, partition of unity)
    g3 = np.load(os.path.join(ROOT, "tests", "golden", "lk_golden3.npz"))
    raw = g3["nn_rows_ssd"]
    out["nn_mi_rows_8"] = np.stack([R.nn_mi_dist_feat(r, 8, False) for r in raw])
    out["nn_mi_rows_10pou"] = np.stack([R.nn_mi_dist_feat(r, 10, True) for r in raw])
    path = os.path.join(ROOT, "tests", "golden", "lk_golden4.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d arrays, %d bytes)" % (path, len(out), os.path.getsize(path)))
    for name, *_ in cases:
        print("  %-18s kept %d of %d, pairs %d" % (name, int(out["fb_%s_mask" % name].sum()), len(out["fb_%s_mask" % name]), len(out["fb_%s_prev_masked" % name])))


if __name__ == "__main__":
    main()
