"""GPU: exact grid kNN statistic behind the EWA variance scale h (rasterizer.py:310-326, 366-388)."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

import scenes
from dss_amd import ops
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
from dss_amd.cloud import PointClouds3D
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_kth(points, K):
    if points.shape[0] == 0:
        return np.zeros(0, np.float32)
    k = min(K, points.shape[0])
    d, _ = cKDTree(points.astype(np.float64)).query(points.astype(np.float64), k=k)
    d = d.reshape(points.shape[0], -1)
    return (d[:, -1] ** 2).astype(np.float32)


@pytest.mark.parametrize("K", [1, 7, 12, 16])
def test_knn_kth_matches_kdtree_multi_cloud(K):
    rng = np.random.default_rng(K)
    bunny, _ = scenes.load_cloud("bunny")
    clouds = [scenes.normalize_unit_sphere(bunny),                      # surface
              rng.uniform(-1, 1, (5000, 3)).astype(np.float32),       # volume
              rng.normal(0, 0.01, (300, 3)).astype(np.float32) + 5,   # tiny far-away cluster
              rng.uniform(0, 1, (3, 3)).astype(np.float32),           # fewer points than K
              np.zeros((0, 3), np.float32),                           # empty
              np.repeat(rng.uniform(0, 1, (40, 3)), 5, 0).astype(np.float32)]  # duplicates
    pts = np.concatenate(clouds, 0)
    num = np.array([c.shape[0] for c in clouds], np.int64)
    first = np.cumsum(num) - num
    got = ops.knn_kth_sqdist(torch.from_numpy(pts).to(DEV), torch.from_numpy(first).to(DEV),
                             torch.from_numpy(num).to(DEV), K).cpu().numpy()
    want = np.concatenate([_ref_kth(c, K) for c in clouds])
    # exact neighbour sets; distances differ only by fp32 vs fp64 evaluation of the same pair
    assert np.allclose(got, want, rtol=2e-5, atol=1e-9), np.abs(got - want).max()


def test_knn_large_cloud_and_global_h():
    pts, nrm, col = scenes.synthetic_cloud(200_000, seed=3)
    t = torch.from_numpy(pts).to(DEV)
    first = torch.zeros(1, dtype=torch.int64, device=DEV)
    num = torch.full((1,), pts.shape[0], dtype=torch.int64, device=DEV)
    got = ops.knn_kth_sqdist(t, first, num, 7).cpu().numpy()
    want = _ref_kth(pts, 7)
    assert np.allclose(got, want, rtol=2e-5, atol=1e-10)
    h = ops.cloud_mean_clamp(torch.from_numpy(got).to(DEV), first, num, 0.5, 5e-5, 1e-3, 0.5e-3, 7).item()
    assert abs(h - float(np.clip((0.5 * want.astype(np.float64)).mean(), 5e-5, 1e-3))) <= 1e-6 * h + 1e-12
    h2 = ops.cloud_mean_clamp(torch.from_numpy(got).to(DEV), first, num, 0.5, 5e-5, 1e-3, 0.5e-3, 7).item()
    assert h == h2  # deterministic


def test_knn_at_the_grid_resolution_cap():
    """500k points per cloud: the per-cloud grid hits its resolution cap (128^3 cells, 2048 scan blocks) and two
    clouds of different sizes share the cell arrays with the stride of the larger one.  K-th distances and full lists
    against the KD-tree on a sample of the queries."""
    from scipy.spatial import cKDTree
    a, _, _ = scenes.synthetic_cloud(500_000, seed=5)
    b, _, _ = scenes.synthetic_cloud(30_000, seed=6)
    pts = np.concatenate([a, b * 0.5 + 0.2]).astype(np.float32)
    first = torch.tensor([0, len(a)], dtype=torch.int64, device=DEV)
    num = torch.tensor([len(a), len(b)], dtype=torch.int64, device=DEV)
    t = torch.from_numpy(pts).to(DEV)
    kth = ops.knn_kth_sqdist(t, first, num, 7).cpu().numpy()
    dists, idx = ops.knn_points(t, first, num, 12)
    dists, idx = dists.cpu().numpy(), idx.cpu().numpy()
    rng = np.random.default_rng(0)
    for f, n in ((0, len(a)), (len(a), len(b))):
        cloud = pts[f: f + n].astype(np.float64)
        sel = rng.choice(n, 5000, replace=False)
        d_ref, i_ref = cKDTree(cloud).query(cloud[sel], k=12)
        assert np.allclose(kth[f + sel], d_ref[:, 6] ** 2, rtol=2e-5, atol=1e-12)
        assert np.allclose(dists[f + sel], d_ref ** 2, rtol=2e-5, atol=1e-12)
        assert (idx[f + sel, 0] == sel).all()
        assert (idx[f + sel] == i_ref).mean() > 0.999        # ties aside, the same neighbours


def test_rasterizer_computes_h_itself():
    """SurfaceSplatting without a precomputed Vrk_h: global (Vrk_invariant) and per-point (isotropic) scales."""
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    R, T = look_at_view_transform(2.0, 30.0, 45.0)
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    cloud = PointClouds3D([torch.from_numpy(pts).to(DEV)], [torch.from_numpy(nrm).to(DEV)],
                          [torch.ones(pts.shape[0], 3, device=DEV)])
    for inv, iso in ((True, False), (False, True)):
        st = PointsRasterizationSettings(backface_culling=False, Vrk_invariant=inv, Vrk_isotropic=iso, image_size=128,
                                         points_per_pixel=5, bin_size=None, radii_backward_scaler=5)
        rast = SurfaceSplatting(cameras=cams, raster_settings=st)
        frags, _ = rast(cloud)
        assert frags.occupancy.mean().item() > 0.05
        if inv:
            assert abs(rast._Vrk_h.item() - scenes.global_h(pts)) <= 1e-5 * scenes.global_h(pts)
        else:
            want = np.clip(0.5 * _ref_kth(pts, 7), 5e-5, 0.01)
            assert np.allclose(rast._Vrk_h.cpu().numpy(), want, rtol=2e-5)


@pytest.mark.parametrize("K", [1, 8, 12, 34])
def test_knn_points_lists_match_kdtree(K):
    """Full neighbour lists (the self query of the regularisers, losses.py:157-180): distances
    to 2e-5 of scipy's fp64 KD-tree, indices identical wherever neighbouring distances are separated (duplicates and
    near-ties may swap), self first, zero padding for clouds smaller than K; deterministic."""
    rng = np.random.default_rng(100 + K)
    bunny, _ = scenes.load_cloud("bunny")
    clouds = [scenes.normalize_unit_sphere(bunny), rng.uniform(-1, 1, (4000, 3)).astype(np.float32),
              rng.uniform(0, 1, (5, 3)).astype(np.float32), np.zeros((0, 3), np.float32),
              rng.normal(0, 0.02, (700, 3)).astype(np.float32) - 3]
    pts = np.concatenate(clouds, 0)
    num = np.array([c.shape[0] for c in clouds], np.int64)
    first = np.cumsum(num) - num
    args = (torch.from_numpy(pts).to(DEV), torch.from_numpy(first).to(DEV), torch.from_numpy(num).to(DEV), K)
    d_t, i_t = ops.knn_points(*args)
    d2_t, i2_t = ops.knn_points(*args)
    assert torch.equal(d_t, d2_t) and torch.equal(i_t, i2_t)
    d, i = d_t.cpu().numpy(), i_t.cpu().numpy()
    assert d.shape == (pts.shape[0], K) and i.dtype == np.int64
    for c, f in zip(clouds, first):
        n = c.shape[0]
        if n == 0:
            continue
        k = min(K, n)
        wd, wi = cKDTree(c.astype(np.float64)).query(c.astype(np.float64), k=k)
        wd, wi = wd.reshape(n, k), wi.reshape(n, k)
        gd, gi = d[f:f + n], i[f:f + n]
        assert np.allclose(gd[:, :k], (wd ** 2).astype(np.float32), rtol=2e-5, atol=1e-9)
        assert (np.diff(gd[:, :k], axis=1) >= 0).all() and (gd[:, 0] == 0).all()
        assert (gi[:, :k] >= 0).all() and (gi[:, :k] < n).all()
        assert (gd[:, k:] == 0).all() and (gi[:, k:] == 0).all()          # padding
        # a listed neighbour really is at the listed distance
        diff = c[gi[:, :k]] - c[:, None, :]
        assert np.allclose((diff.astype(np.float64) ** 2).sum(-1), gd[:, :k], rtol=2e-5, atol=1e-9)
        # indices agree with the KD-tree wherever the ordering is unambiguous
        sep = np.ones((n, k), bool)
        if k > 1:
            gap = np.diff(wd, axis=1) > 1e-6 * np.maximum(wd[:, 1:], 1e-12)
            sep[:, 1:] &= gap
            sep[:, :-1] &= gap
        assert (gi[:, :k][sep] == wi[sep]).all()


def _ref_radius_stat(points, K, r):
    """frnn_grid_points(K, r) followed by `sq_dist[:, :, 1:].max(-1)` (rasterizer.py:317-324): neighbours beyond r come back
    as -1; the statistic is the farthest of the K - 1 nearest non-self neighbours that lies within r, -1 if there is none"""
    P = points.shape[0]
    if P == 0:
        return np.zeros(0, np.float32)
    k = min(K, P)
    d, _ = cKDTree(points.astype(np.float64)).query(points.astype(np.float64), k=k)
    d2 = d.reshape(P, -1)[:, 1:] ** 2
    d2 = np.where(d2 <= float(r) ** 2, d2, -1.0)
    return (d2.max(1) if d2.shape[1] else np.full(P, -1.0)).astype(np.float32)


@pytest.mark.parametrize("r", [0.2, 0.05])
def test_knn_fixed_radius_statistic_of_the_references_default_search(r):
    """round 6: SurfaceSplatting's default frnn_radius = 0.2 selects frnn_grid_points(K = 7, r), whose
    missing neighbours count as -1 -- an isolated point contributes -0.5 to the cloud's mean h."""
    rng = np.random.default_rng(11)
    bunny, _ = scenes.load_cloud("bunny")
    surface = scenes.normalize_unit_sphere(bunny)[::3]
    strays = rng.uniform(-3, 3, (60, 3)).astype(np.float32)          # far from everything: no neighbour within r
    pairs = np.repeat(rng.uniform(4, 5, (10, 3)), 2, 0).astype(np.float32) + rng.normal(0, 0.01, (20, 3)).astype(np.float32)
    clouds = [np.concatenate([surface, strays, pairs]), rng.uniform(-1, 1, (4000, 3)).astype(np.float32)]
    pts = np.concatenate(clouds, 0)
    num = np.array([c.shape[0] for c in clouds], np.int64)
    first = np.cumsum(num) - num
    t = lambda a: torch.from_numpy(a).to(DEV)
    got = ops.knn_kth_sqdist(t(pts), t(first), t(num), 7, radius=r).cpu().numpy()
    want = np.concatenate([_ref_radius_stat(c, 7, r) for c in clouds])
    assert ((got < 0) == (want < 0)).all() and (want < 0).sum() >= 40
    assert np.allclose(got, want, rtol=2e-5, atol=1e-9), np.abs(got - want).max()
    # radius None / <= 0: the plain K-th distance
    assert torch.equal(ops.knn_kth_sqdist(t(pts), t(first), t(num), 7, radius=-1.0), ops.knn_kth_sqdist(t(pts), t(first), t(num), 7))


@pytest.mark.parametrize("culling", ["all", "none", "mixed"])
@pytest.mark.parametrize("shared", [True, False])
def test_knn_statistic_in_the_references_order_under_depth_culling(shared, culling):
    """The reference drops, per camera, the points outside [znear, zfar] BEFORE its neighbour search and takes the mean over
    the padded batch (rasterizer.py:599, 183-217, 310-326): `knn_kth_sqdist_view` + `renderable_mean_clamp` against a KD-tree
    over each camera's kept subset."""
    rng = np.random.default_rng(5)
    bunny, _ = scenes.load_cloud("bunny")
    base = scenes.normalize_unit_sphere(bunny)[::2].astype(np.float32)
    Mn, Vn, _ = scenes.camera_matrices([1.3, 1.6, 2.5], [10.0, 40.0, -20.0], [0.0, 120.0, 250.0])
    N = Vn.shape[0]
    znear, zfar = np.array([1.0, 1.2, 1.0], np.float32), np.array([100.0, 1.9, 2.6], np.float32)
    # "none": no camera drops a point (every row is the plain statistic: ONE search, the other rows are copies);
    # "mixed": only the second camera does (its own masked search; the other two share the plain one)
    if culling == "none":
        znear, zfar = np.full(3, 0.01, np.float32), np.full(3, 100.0, np.float32)
    elif culling == "mixed":
        znear, zfar = np.array([0.01, 1.2, 0.01], np.float32), np.array([100.0, 1.9, 100.0], np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    clouds = [base] if shared else [base, base[::2] * 1.1, base[1::3]]
    pts = np.concatenate(clouds, 0)
    num = np.array([c.shape[0] for c in clouds], np.int64)
    first = np.cumsum(num) - num
    r = 0.2
    got = ops.knn_kth_sqdist_view(t(pts), t(first), t(num), 7, t(Vn), t(znear), t(zfar), shared, radius=r).cpu().numpy()
    sums, cnts = [], []
    for n in range(N):
        c = clouds[0] if shared else clouds[n]
        z = c[:, 0] * Vn[n, 0, 2] + c[:, 1] * Vn[n, 1, 2] + c[:, 2] * Vn[n, 2, 2] + Vn[n, 3, 2]
        ok = (z >= znear[n]) & (z <= zfar[n])
        drops = culling == "all" or (culling == "mixed" and n == 1)
        assert (0 < ok.sum() < c.shape[0]) if drops else bool(ok.all())
        want = _ref_radius_stat(c[ok], 7, r)
        mine = got[n] if shared else got[first[n]:first[n] + num[n]]
        assert np.allclose(mine[ok], want, rtol=2e-5, atol=1e-9), (n, np.abs(mine[ok] - want).max())
        assert (mine[~ok] == 0).all()
        sums.append(float((0.5 * want.astype(np.float64)).sum()))
        cnts.append(int(ok.sum()))
    f1 = t(np.zeros(N, np.int64)) if shared else t(first)
    n1 = t(np.full(N, num[0], np.int64)) if shared else t(num)
    h = ops.renderable_mean_clamp(t(got), t(pts), t(Vn), t(znear), t(zfar), f1, n1, shared, 0.5, 5e-5, 1e-3, 0.5e-3, 7).cpu().numpy()
    want_h = np.clip(np.array(sums) / max(cnts), 5e-5, 1e-3)   # mean over the PADDED batch: the largest kept count
    assert np.allclose(h, want_h, rtol=1e-5), (h, want_h)


@pytest.mark.parametrize("mode", ["invariant", "isotropic"])
def test_masked_culling_renders_what_the_references_drop_then_search_order_renders(mode):
    """Round 6: with cameras that cull DIFFERENT points of one shared cloud, the sync-free masked path (culled points keep their
    slot) must give the image of the reference's order -- extend, drop per camera, THEN search the neighbours and derive h
    (`compact_culled=True` does literally that, with host syncs): the variance scale is per camera (invariant: mean over the
    padded batch) or per (camera, point) pair (isotropic), from the fixed-radius search of the default frnn_radius = 0.2."""
    from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
    bunny, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(bunny)
    rng = np.random.default_rng(2)
    # a few strays (no neighbour within 0.2) and a cloud deep enough that tight depth ranges cut it differently per camera
    strays = rng.uniform(-1.5, 1.5, (2, 3)).astype(np.float32)   # (each pulls the mean h down by 0.5 / P: two keep it unclamped)
    pts = np.concatenate([pts, strays]).astype(np.float32)
    nrm = np.concatenate([nrm, np.tile(np.array([[0, 0, 1]], np.float32), (2, 1))]).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    R, T = look_at_view_transform([2.0, 2.2], [20.0, -10.0], [30.0, 200.0])
    cams = FoVPerspectiveCameras(fov=60.0, R=R, T=T, device=DEV)
    cams.znear = torch.tensor([1.9, 1.0], device=DEV)
    cams.zfar = torch.tensor([100.0, 2.3], device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, depth_merging_threshold=0.05,
                                     Vrk_invariant=mode == "invariant", Vrk_isotropic=mode == "isotropic",
                                     radii_backward_scaler=5.0, image_size=128, points_per_pixel=5, bin_size=None,
                                     clip_pts_grad=0.05, antialiasing_sigma=1.0)
    col = torch.rand((pts.shape[0], 3), generator=torch.Generator().manual_seed(1)).to(DEV)
    images, hs = {}, {}
    for compact in (False, True):
        ras = SurfaceSplatting(cameras=cams, raster_settings=st, compact_culled=compact)
        ren = SurfaceSplattingRenderer(ras, NormWeightedCompositor())
        with torch.no_grad():
            images[compact] = ren(PointClouds3D([t(pts)], [t(nrm)], [col]))
        hs[compact] = ras._Vrk_h
        if not compact:
            assert hs[compact].numel() == (2 if mode == "invariant" else 2 * pts.shape[0])
    if mode == "invariant":   # one h per camera, the same in both orders, not clamped away, different per camera
        assert torch.allclose(hs[False], hs[True], rtol=1e-6), (hs[False], hs[True])
        assert 5e-5 < float(hs[False].min()) and float(hs[False].max()) < 1e-3 and float(hs[False][0]) != float(hs[False][1]), hs[False]
    a, b = images[False], images[True]
    assert a.shape == b.shape and float(a[..., 3].sum()) > 500
    assert torch.equal(a[..., 3], b[..., 3]) and float((a - b).abs().max()) <= 1e-6, float((a - b).abs().max())


def _clustered_clouds():
    """-> list of clouds: [0] the model of the reference's training loop after 893 iterations at configs[2] (dense bulk, thin
    halo out to radius 2.1: tests/golden/trained_cloud_cfg3.npz), [1] a synthetic one with a cell of more than 4,096 points
    (not sub-sorted), exact duplicates and far outliers, [2] an evenly sampled surface (no dense cell), [3] three points, [4] none."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_cloud_cfg3.npz"))
    rng = np.random.default_rng(7)
    blob = rng.normal(0, 0.004, (12000, 3)).astype(np.float32)                      # > 4096 points in one cell
    shell = rng.normal(0, 1, (30000, 3)); shell = (0.4 * shell / np.linalg.norm(shell, axis=1, keepdims=True)).astype(np.float32)
    clump = (rng.normal(0, 0.02, (20000, 3)) + [0.3, 0.1, -0.2]).astype(np.float32)
    dup = np.repeat(rng.uniform(-0.05, 0.05, (500, 3)), 4, 0).astype(np.float32)
    far = rng.uniform(-3, 3, (300, 3)).astype(np.float32)
    synth = np.concatenate([blob, shell, clump, dup, far])[rng.permutation(12000 + 30000 + 20000 + 2000 + 300)]
    even, _, _ = scenes.synthetic_cloud(20000, seed=9)
    tiny = rng.uniform(-0.3, 0.3, (3, 3)).astype(np.float32)      # fewer points than K: the shortcut of the per-camera search must not fire
    return [z["points"].astype(np.float32), synth, even.astype(np.float32), tiny, np.zeros((0, 3), np.float32)]


def _with_query_option(value, fn):
    from dss_amd import _lib
    _lib.set_option(_lib.OPT_KNN_QUERY, value)
    try:
        return fn()
    finally:
        _lib.set_option(_lib.OPT_KNN_QUERY, 0)


@pytest.mark.parametrize("which", ["trained", "packed"])
def test_knn_skip_structure_of_clustered_clouds_is_exact(which):
    """From 65,536 points on, cells with more than 64 points are ordered along a Morton curve and long candidate runs are walked
    in 16-slot blocks whose boxes are tested against the current K-th distance (knn.hip, knn_subsort_kernel).  Same results,
    bit for bit, as the uniform walk (DSS_OPT_KNN_QUERY = 3), which the tests above pin against the KD-tree -- K-th distance,
    full (distance, id) lists with ties, fixed-radius statistic, per-camera culling -- and against the KD-tree directly."""
    clouds = _clustered_clouds()
    clouds = clouds[:1] if which == "trained" else clouds
    pts = np.concatenate(clouds, 0)
    num = np.array([c.shape[0] for c in clouds], np.int64)
    first = np.cumsum(num) - num
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    P, F, Nn = t(pts), t(first), t(num)
    # K-th distance, plain and fixed-radius
    for K, r in ((7, None), (7, 0.2), (16, None)):
        a = ops.knn_kth_sqdist(P, F, Nn, K, radius=r)
        b = _with_query_option(3, lambda: ops.knn_kth_sqdist(P, F, Nn, K, radius=r))
        assert torch.equal(a, b), (K, r, (a != b).sum().item())
        want = np.concatenate([_ref_kth(c, K) if r is None else _ref_radius_stat(c, K, r) for c in clouds])
        assert np.allclose(a.cpu().numpy(), want, rtol=2e-5, atol=1e-9), np.abs(a.cpu().numpy() - want).max()
    # full lists
    for K in (12, 34):
        da, ia = ops.knn_points(P, F, Nn, K)
        db, ib = _with_query_option(3, lambda: ops.knn_points(P, F, Nn, K))
        assert torch.equal(da, db) and torch.equal(ia, ib), K
    rng = np.random.default_rng(0)
    da, ia = ops.knn_points(P, F, Nn, 12)
    da, ia = da.cpu().numpy(), ia.cpu().numpy()
    for f, n, c in zip(first, num, clouds):
        if n < 4000:
            continue
        sel = rng.choice(n, 4000, replace=False)
        d_ref, i_ref = cKDTree(c.astype(np.float64)).query(c[sel].astype(np.float64), k=12)
        assert np.allclose(da[f + sel], d_ref ** 2, rtol=2e-5, atol=1e-12)
        clear = (np.diff(d_ref, axis=1) > 1e-6 * d_ref[:, 1:]).all(1)   # rows without (near-)ties: the ids are determined
        assert clear.sum() > 1000 and (ia[f + sel][clear] == i_ref[clear]).all()
    # per-camera culling (one shared cloud seen by three cameras; one cloud per camera)
    nc = 3 if which == "trained" else len(clouds)
    Mn, Vn, _ = scenes.camera_matrices([1.3, 1.6, 2.5, 1.4, 1.4][:nc], [10.0, 40.0, -20.0, 5.0, 5.0][:nc], [0.0, 120.0, 250.0, 60.0, 60.0][:nc])
    znear = np.array([1.0, 0.01, 1.0, 1.35, 1.0], np.float32)[:nc]
    zfar = np.array([100.0, 100.0, 2.6, 100.0, 100.0], np.float32)[:nc]
    args = (P, F, Nn, 7, t(Vn), t(znear), t(zfar), which == "trained")
    for r in (0.2, None):
        a = ops.knn_kth_sqdist_view(*args, radius=r)
        b = _with_query_option(3, lambda: ops.knn_kth_sqdist_view(*args, radius=r))
        assert torch.equal(a, b), r
    a = ops.knn_kth_sqdist_view(*args, radius=0.2)
    if which != "trained":
        # the three-point cloud under a camera that drops part of it: the statistic among the points it keeps
        c3, f3 = clouds[3], int(first[3])
        zv3 = c3[:, 0] * Vn[3, 0, 2] + c3[:, 1] * Vn[3, 1, 2] + c3[:, 2] * Vn[3, 2, 2] + Vn[3, 3, 2]
        ok3 = (zv3 >= znear[3]) & (zv3 <= zfar[3])
        got3 = a[f3:f3 + 3].cpu().numpy()
        assert (got3[~ok3] == 0).all() and np.allclose(got3[ok3], _ref_radius_stat(c3[ok3], 7, 0.2), rtol=2e-5, atol=1e-9), (ok3, got3)
    c0 = clouds[0]
    zv = c0[:, 0] * Vn[0, 0, 2] + c0[:, 1] * Vn[0, 1, 2] + c0[:, 2] * Vn[0, 2, 2] + Vn[0, 3, 2]
    ok = (zv >= znear[0]) & (zv <= zfar[0])
    assert 0 < ok.sum() < len(c0)
    mine = (a[0] if which == "trained" else a[:len(c0)]).cpu().numpy()
    assert np.allclose(mine[ok], _ref_radius_stat(c0[ok], 7, 0.2), rtol=2e-5, atol=1e-9) and (mine[~ok] == 0).all()


@pytest.mark.parametrize("mode", ["invariant", "isotropic"])
def test_cloned_copies_of_one_cloud_are_searched_once(mode):
    """The reference's texture hands ONE model cloud over as N clones (`pointclouds.extend(N)`, texture.py:88): separate tensors,
    same positions.  `SurfaceSplatting` compares the copies (one launch, one host read) and searches the neighbour statistic
    in the first one, as a cloud shared by the N cameras with each camera's own culling -- same h, bit for bit, as the N-cloud
    search (`detect_identical_clouds=False`), and the same image; copies that differ in one coordinate take the N-cloud search."""
    bunny, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(bunny)[::2].astype(np.float32)
    nrm = nrm[::2].astype(np.float32)
    R, T = look_at_view_transform([1.3, 1.6, 2.5], [10.0, 40.0, -20.0], [0.0, 120.0, 250.0])
    cams = FoVPerspectiveCameras(znear=torch.tensor([1.0, 0.01, 1.0]), zfar=torch.tensor([100.0, 100.0, 2.6]), R=R, T=T, device=DEV)
    N = 3
    t = lambda a: torch.from_numpy(a).to(DEV)
    st = PointsRasterizationSettings(backface_culling=False, Vrk_invariant=mode == "invariant", Vrk_isotropic=mode == "isotropic",
                                     image_size=96, points_per_pixel=5, bin_size=None, radii_backward_scaler=5)

    def run(points_list, detect):
        cloud = PointClouds3D(points_list, [t(nrm) for _ in range(N)], [torch.ones(pts.shape[0], 3, device=DEV) for _ in range(N)])
        rast = SurfaceSplatting(cameras=cams, raster_settings=st, detect_identical_clouds=detect)
        frags, _ = rast(cloud)
        return rast._Vrk_h.clone(), frags

    h_once, f_once = run([t(pts).clone() for _ in range(N)], True)
    h_each, f_each = run([t(pts).clone() for _ in range(N)], False)
    assert h_once.shape == h_each.shape and torch.equal(h_once, h_each)
    assert torch.equal(f_once.idx, f_each.idx) and torch.equal(f_once.zbuf, f_each.zbuf)
    # a copy that differs: the general search, and a different h for that camera's cloud
    moved = pts.copy()
    moved[::2] *= 1.2      # (the third camera's h is not at its clamp: the statistic of ITS cloud must show)
    h_diff, _ = run([t(pts).clone(), t(pts).clone(), t(moved)], True)
    h_diff_each, _ = run([t(pts).clone(), t(pts).clone(), t(moved)], False)
    assert torch.equal(h_diff, h_diff_each) and not torch.equal(h_diff, h_once)
