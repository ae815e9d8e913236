#!/usr/bin/env python3
"""Developer tool (GPU box): randomised self-consistency sweep of the two structures round 6 added for clustered inputs.
  * rasterizer: the binned forward (tile lists, spill pool, 64-bit masks and block records of wide splats) against the
    whole-cloud scan of the same library (`bin_size=0`: no lists at all) -- fragments must be identical;
  * neighbour search: the skip structure against the uniform walk (DSS_OPT_KNN_QUERY = 3) -- K-th distances, (distance, id)
    lists and the per-camera statistic must be identical.
    python tools/fuzz_clustered.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from dss_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
bad = 0
for c in range(cases):
    # ---- rasterizer
    S = int(rng.choice([40, 64, 100, 128, 200]))
    N = int(rng.integers(1, 4))
    P = int(rng.integers(200, 9000))
    rmin = float(rng.uniform(0.5, 12.0))
    rmax = rmin + float(rng.uniform(0.5, 60.0))
    K = int(rng.choice([1, 3, 5, 8, 12]))
    thr = float(rng.choice([0.02, 0.05, 0.5, 10.0]))
    sc = scenes.random_splats(P, S, N, seed=int(rng.integers(1 << 30)), rmin=rmin, rmax=rmax, ties=bool(rng.integers(2)))
    d = dict(points=t(sc["points"]), ellipse=t(sc["ellipse"]), cutoff=t(sc["cutoff"]), radii=t(sc["radii"]),
             first=t(sc["first_idx"]), num=t(sc["num_pts"]))
    a = ops.splat_points(d["points"], d["ellipse"], d["cutoff"], d["radii"], d["first"], d["num"], thr, S, K, None, None)
    b = ops.splat_points(d["points"], d["ellipse"], d["cutoff"], d["radii"], d["first"], d["num"], thr, S, K, 0, None)
    ok = all(torch.equal(x, y) for x, y in zip(a, b))
    bad += not ok
    print("raster case %2d S=%3d N=%d P=%5d r=%.1f..%.1f px K=%2d thr=%g -> %s" % (c, S, N, P, rmin, rmax, K, thr, "ok" if ok else "MISMATCH"), flush=True)
    # ---- neighbour search
    Pn = int(rng.integers(66000, 260000))
    parts, left = [], Pn
    while left > 0:
        m = int(min(left, rng.integers(500, 60000)))
        kind = rng.integers(3)
        centre = rng.uniform(-1, 1, 3)
        if kind == 0:
            parts.append(rng.normal(0, 10 ** rng.uniform(-3, -0.5), (m, 3)) + centre)
        elif kind == 1:
            v = rng.normal(0, 1, (m, 3)); parts.append(centre + rng.uniform(0.05, 0.8) * v / np.linalg.norm(v, axis=1, keepdims=True))
        else:
            parts.append(rng.uniform(-3, 3, (m, 3)))
        left -= m
    pts = np.concatenate(parts).astype(np.float32)[rng.permutation(Pn)]
    split = int(rng.integers(0, 3))
    num = np.array([Pn] if split == 0 else [Pn // 3, Pn - Pn // 3] if split == 1 else [Pn - 7, 4, 3], np.int64)
    first = np.cumsum(num) - num
    Pt, F, Nn = t(pts), t(first), t(num)
    res = []
    for opt in (0, 3):
        _lib.set_option(_lib.OPT_KNN_QUERY, opt)
        kth = ops.knn_kth_sqdist(Pt, F, Nn, 7, radius=0.2 if c % 2 else None)
        dd, ii = ops.knn_points(Pt, F, Nn, int(rng.choice([8, 12, 16])) if opt == 0 else dd.shape[1])
        Mn, Vn, _ = scenes.camera_matrices([1.3, 1.6, 2.5][:len(num)], [10.0, 40.0, -20.0][:len(num)], [0.0, 120.0, 250.0][:len(num)])
        zn = t(np.array([1.0, 0.01, 1.2], np.float32)[:len(num)]); zf = t(np.array([100.0, 100.0, 2.6], np.float32)[:len(num)])
        view = ops.knn_kth_sqdist_view(Pt, F, Nn, 7, t(Vn), zn, zf, False, radius=0.2)
        res.append((kth, dd, ii, view))
    _lib.set_option(_lib.OPT_KNN_QUERY, 0)
    ok = all(torch.equal(x, y) for x, y in zip(res[0], res[1]))
    bad += not ok
    print("knn    case %2d P=%6d clouds=%s K=%d -> %s" % (c, Pn, num.tolist(), res[0][1].shape[1], "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
