#!/usr/bin/env python
"""Generates tests/golden/lk_golden3.npz (r05) from the independent NumPy re-derivation (oracle/numpy_ref.py): the paths the r04
verdict still found covered only by the oracle and the product -- GridTracker's patch layout in its three modes on a region that is
not a parallelogram (SM/src/GridTracker.cc:86-94, 345-380), NN dataset rows with the reference's perturb -> sample -> un-perturb
sequence (SM/src/NT/NN.cc:131-191; SSD and NCC features), and the stochastic samplers for given normals (Homography corner based
sampling Homography.cc:899-909, Affine point based sampling 1 / 2 Affine.cc:464-494).
PARITY UNPINNED with respect to the reference itself (it ships no vectors and cannot be built here).

Run from the repo root:  python tests/golden/make_golden3.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy_ref as R  # noqa: E402
from mtf_amd import synth  # noqa: E402

SEED = 20260930
IMG_SEED = 101
IMG_SHAPE = (224, 224)


def main():
    rng = np.random.default_rng(SEED)
    img = synth.make_frame(*IMG_SHAPE, seed=IMG_SEED)
    out = {"img_seed": IMG_SEED, "img_shape": np.array(IMG_SHAPE)}

    # ---- GridTracker's patch layout: a strongly projective quadrilateral, 5 x 4 patches of 12 x 9
    region = np.array([[40.0, 190.0, 170.0, 55.0], [35.0, 25.0, 185.0, 160.0]])
    out["grid_region"] = region
    out["grid_dims"] = np.array([5, 4, 12, 9])
    for name, dyn, inside in (("inside", 0, 1), ("points", 0, 0), ("dyn", 1, 0)):
        pts, patches = R.grid_layout(region, 5, 4, 12, 9, dyn, inside)
        out["grid_pts_" + name] = pts
        out["grid_patches_" + name] = patches

    # ---- NN dataset rows: 24 x 24 homography template, 12 perturbations
    res = 24
    corners = synth.square_corners(112, 108, 90) + rng.uniform(-2, 2, size=(2, 4))
    _, init_hm = R.grid_from_corners(corners, res, res)
    perts = rng.normal(0, 1, size=(12, 8)) * np.array([0.02, 0.02, 2.0, 0.02, 0.02, 2.0, 1e-4, 1e-4])
    out.update({"nn_corners": corners, "nn_perts": perts, "nn_rows_ssd": R.nn_dataset_rows(img, init_hm, perts, ncc=False),
                "nn_rows_ncc": R.nn_dataset_rows(img, init_hm, perts, ncc=True)})

    # ---- samplers with given draws
    hc = synth.square_corners(110, 100, 70) + rng.uniform(-3, 3, size=(2, 4))
    z10 = rng.normal(size=(6, 10))
    sig_h, mean_h = np.array([1.5, 0.8, 1, 1, 1, 1, 1, 1.0]), np.array([0.2, -0.1, 0, 0, 0, 0, 0, 0.0])
    out.update({"smp_hom_corners": hc, "smp_hom_z": z10, "smp_hom_sigma": sig_h, "smp_hom_mean": mean_h,
                "smp_hom_states": np.stack([R.hom_corner_sampler(hc, sig_h, mean_h, z) for z in z10])})
    ac = synth.square_corners(105, 115, 64) + rng.uniform(-3, 3, size=(2, 4))
    z8 = rng.normal(size=(6, 8))
    sig_a, mean_a = np.array([1.2, 0.7, 0.9, 1.1, 0.6, 0.8, 1, 1.0]), np.array([0.1, -0.2, 0.05, 0.0, -0.1, 0.15, 0, 0.0])
    out.update({"smp_aff_corners": ac, "smp_aff_z": z8, "smp_aff_sigma": sig_a, "smp_aff_mean": mean_a,
                "smp_aff_states_1": np.stack([R.aff_point_sampler(ac, 1, sig_a, mean_a, z) for z in z8]),
                "smp_aff_states_2": np.stack([R.aff_point_sampler(ac, 2, sig_a, mean_a, z) for z in z8])})

    path = os.path.join(ROOT, "tests", "golden", "lk_golden3.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%d arrays, %d bytes)" % (path, len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
