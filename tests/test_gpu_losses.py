"""GPU: the point-cloud regularisers (projection / repulsion, reference DSS/training/losses.py:145-459) through the
C ABI, against (a) the golden vectors produced by the reference's own classes with autograd
(tests/golden/make_golden_losses.py) and (b) the oracle on larger clouds.  Float work: tolerances stated per check."""
import os
import types

import numpy as np
import pytest
import torch

import oracle
import scenes
from dss_amd import ops
from dss_amd.cloud import PointClouds3D
from dss_amd.losses import ProjectionLoss, RepulsionLoss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_losses.npz")


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _golden_clouds(z):
    pts = [z["points_a"], z["points_b"]]
    nrm = [z["normals_a"], z["normals_b"]]
    num = np.array([len(p) for p in pts], np.int64)
    first = np.concatenate([[0], np.cumsum(num)[:-1]]).astype(np.int64)
    return pts, nrm, first, num


@pytest.mark.parametrize("knn_k", [12, 33])
def test_kernels_match_reference_golden(knn_k):
    """dss_knn_points -> dss_mollify_normals -> dss_projection_loss / dss_repulsion_loss vs the reference run.
    Tolerances: mollified normals 1e-4 relative; losses rtol 2e-3 (fp32 cancellation in the plane distances);
    gradients rel-L2 <= 1e-3 (north-star gradient bar)."""
    z = np.load(GOLD)
    tag = "k%d" % knn_k
    _, sigma, fscale = (float(v) for v in z[tag + "_params"])
    pts, nrm, first, num = _golden_clouds(z)
    P = _t(np.concatenate(pts)); Nn = _t(np.concatenate(nrm)); F = _t(first); L = _t(num)
    dists, idx = ops.knn_points(P, F, L, knn_k)
    keep = _t(z["visibility"] & z["inmask"])
    moll = ops.mollify_normals(Nn, dists, idx, keep, F, L)
    assert np.allclose(moll.cpu().numpy(), z[tag + "_mollified"], rtol=1e-4, atol=1e-6)

    loss, grad = ops.projection_loss(P, moll, dists, idx, _t(z["visibility"]), F, L, sigma,
                                     grad_loss=_t(z[tag + "_proj_gup"]), want_grad=True)
    assert np.allclose(loss.cpu().numpy(), z[tag + "_proj_loss"], rtol=2e-3, atol=1e-9)
    assert _rel(grad.cpu().numpy(), z[tag + "_proj_grad"]) <= 1e-3

    lossr, gradr = ops.repulsion_loss(P, moll, idx, F, L, sigma, fscale, grad_loss=_t(z[tag + "_repel_gup"]), want_grad=True)
    assert np.allclose(lossr.cpu().numpy(), z[tag + "_repel_loss"], rtol=1e-4, atol=1e-6)
    assert _rel(gradr.cpu().numpy(), z[tag + "_repel_grad"]) <= 1e-3


def test_loss_modules_match_reference_golden():
    """The drop-in classes (same constructor / call as trainer.py:134-137, 319-326): mean-reduced value and the
    gradient autograd delivers to the point parameters."""
    z = np.load(GOLD)
    pts, nrm, first, num = _golden_clouds(z)
    params = [torch.nn.Parameter(_t(p)) for p in pts]
    pc = PointClouds3D(params, [_t(n) for n in nrm])
    maxp = int(num.max())
    vis = torch.zeros(2, maxp, dtype=torch.bool, device=DEV)
    inm = torch.zeros(2, maxp, dtype=torch.bool, device=DEV)
    for b in range(2):
        vis[b, : num[b]] = _t(z["visibility"][first[b]: first[b] + num[b]])
        inm[b, : num[b]] = _t(z["inmask"][first[b]: first[b] + num[b]])
    flt = types.SimpleNamespace(visibility=vis, inmask=inm)   # padded (N, Pmax) masks like PointCloudsFilters

    proj = ProjectionLoss(reduction="mean", filter_scale=2.0, knn_k=12)
    val = proj(pc, rebuild_knn=True, points_filter=flt)
    assert abs(val.item() - float(z["k12_proj_mean"])) <= 2e-4 * float(z["k12_proj_mean"])
    # gradient of sum(loss * g_up): the reference vectors were produced with that upstream gradient
    per_point = proj(pc, rebuild_knn=False, points_filter=flt, reduction="none")
    (per_point * _t(z["k12_proj_gup"])).sum().backward()
    g = torch.cat([p.grad for p in params]).cpu().numpy()
    assert _rel(g, z["k12_proj_grad"]) <= 1e-3

    for p in params:
        p.grad = None
    rep = RepulsionLoss(reduction="none", filter_scale=2.0, knn_k=12)
    lr = rep(pc, rebuild_knn=True, points_filter=flt)
    assert tuple(lr.shape) == (int(num.sum()), 3)
    assert np.allclose(lr.detach().cpu().numpy(), z["k12_repel_loss"], rtol=1e-4, atol=1e-6)
    (lr * _t(z["k12_repel_gup"])).sum().backward()
    g = torch.cat([p.grad for p in params]).cpu().numpy()
    assert _rel(g, z["k12_repel_grad"]) <= 1e-3


@pytest.mark.parametrize("with_masks", [False, True])
def test_kernels_match_oracle_on_bunny(with_masks):
    """32,684-point cfg2 cloud (+ a second, smaller cloud) against the double-precision oracle."""
    rng = np.random.default_rng(5)
    pts, nrm = scenes.load_cloud("bunny")
    pts, nrm = scenes.upsample_jitter(scenes.normalize_unit_sphere(pts), nrm, 4, seed=0)
    sel = rng.permutation(len(pts))[:9000]
    clouds = [pts.astype(np.float32), (pts[sel] * 0.7 + 0.1).astype(np.float32)]
    normals = [(nrm * rng.uniform(0.5, 1.5, (len(nrm), 1)) + rng.normal(0, 0.2, nrm.shape)).astype(np.float32),
               nrm[sel].astype(np.float32)]
    num = np.array([len(c) for c in clouds], np.int64)
    first = np.array([0, num[0]], np.int64)
    Pn, Nn = np.concatenate(clouds), np.concatenate(normals)
    P, Nt, F, L = _t(Pn), _t(Nn), _t(first), _t(num)
    K = 12
    dists, idx = ops.knn_points(P, F, L, K)
    d_np, i_np = dists.cpu().numpy(), idx.cpu().numpy()
    first_of = np.repeat(first, num)
    vis = rng.random(len(Pn)) < 0.5 if with_masks else None
    keep = (vis & (rng.random(len(Pn)) < 0.7)) if with_masks else None

    moll = ops.mollify_normals(Nt, dists, idx, None if keep is None else _t(keep), F, L)
    moll_o = oracle.mollify_normals(Nn, d_np, i_np, keep, first_of)
    assert np.allclose(moll.cpu().numpy(), moll_o, rtol=1e-4, atol=1e-6)

    g1 = rng.normal(0, 1, len(Pn)).astype(np.float32)
    loss, grad = ops.projection_loss(P, _t(moll_o), dists, idx, None if vis is None else _t(vis), F, L, 0.75,
                                     grad_loss=_t(g1), want_grad=True)
    loss_o, grad_o = oracle.projection_loss(Pn, moll_o, d_np, i_np, vis, first_of, 0.75, g1)
    assert np.allclose(loss.cpu().numpy(), loss_o, rtol=2e-3, atol=1e-10)
    assert _rel(grad.cpu().numpy(), grad_o) <= 1e-3

    g3 = rng.normal(0, 1, (len(Pn), 3)).astype(np.float32)
    lossr, gradr = ops.repulsion_loss(P, _t(moll_o), idx, F, L, 0.75, 2.0, grad_loss=_t(g3), want_grad=True)
    inv_of = np.concatenate([np.full(len(c), np.float32(len(c)) / np.float32(((c.max(0) - c.min(0)) ** 2).sum()) * 2.0,
                                     np.float32) for c in clouds])
    lossr_o, gradr_o = oracle.repulsion_loss(Pn, moll_o, i_np, first_of, inv_of, 0.75, g3)
    assert np.allclose(lossr.cpu().numpy(), lossr_o, rtol=1e-4, atol=1e-6)
    assert _rel(gradr.cpu().numpy(), gradr_o) <= 1e-3
    # forward-only and backward-only calls write the same values as the combined call
    only_l, none_g = ops.repulsion_loss(P, _t(moll_o), idx, F, L, 0.75, 2.0)
    assert none_g is None and torch.equal(only_l, lossr)
    none_l, only_g = ops.projection_loss(P, _t(moll_o), dists, idx, None if vis is None else _t(vis), F, L, 0.75,
                                         grad_loss=_t(g1), want_loss=False, want_grad=True)
    assert none_l is None and torch.equal(only_g, grad)


def test_projection_gradient_is_the_derivative_of_the_loss():
    """With the weights frozen (as the reference freezes them) loss_i is a quadratic in p_i: a central difference on
    the oracle, moving only points that are not neighbours of one another (so every moved point still sees its
    neighbours at their original positions), reproduces the analytic gradient."""
    rng = np.random.default_rng(2)
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)[:3000].astype(np.float32)
    nrm = nrm[:3000].astype(np.float32)
    F = _t(np.array([0], np.int64)); L = _t(np.array([len(pts)], np.int64))
    P, Nt = _t(pts), _t(nrm)
    dists, idx = ops.knn_points(P, F, L, 12)
    moll = ops.mollify_normals(Nt, dists, idx, None, F, L)
    _, grad = ops.projection_loss(P, moll, dists, idx, None, F, L, 0.75, want_grad=True)
    d_np, i_np, moll_np = dists.cpu().numpy(), idx.cpu().numpy(), moll.cpu().numpy()
    chosen, touched = [], set()
    for p in rng.permutation(len(pts)):
        nbrs = set(i_np[p].tolist())
        if not (nbrs & touched) and p not in touched:
            chosen.append(p)
            touched |= nbrs   # neither a chosen point nor any of its neighbours may be chosen again
    chosen = np.array(chosen)
    assert len(chosen) > 100
    d = rng.normal(0, 1, (len(chosen), 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    eps = 1e-3
    first_of = np.zeros(len(pts), np.int64)
    vals = []
    for sign in (+1, -1):
        q = pts.astype(np.float64).copy()
        q[chosen] += sign * eps * d
        vals.append(oracle.projection_loss(q, moll_np, d_np, i_np, None, first_of, 0.75)[0][chosen].astype(np.float64))
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = (grad.cpu().numpy()[chosen].astype(np.float64) * d).sum(1)
    assert np.linalg.norm(fd - an) <= 2e-2 * np.linalg.norm(an)   # fp32 loss values limit the difference quotient


def test_bad_arguments_fail_loudly():
    P = torch.rand(100, 3, device=DEV)
    F = torch.zeros(1, dtype=torch.int64, device=DEV); L = torch.full((1,), 100, dtype=torch.int64, device=DEV)
    dists, idx = ops.knn_points(P, F, L, 8)
    with pytest.raises(RuntimeError, match="K"):
        ops.mollify_normals(P, dists[:, :1].contiguous(), idx[:, :1].contiguous(), None, F, L)
    with pytest.raises(RuntimeError, match="one entry per packed point"):
        ops.mollify_normals(P, dists, idx, torch.ones(7, dtype=torch.bool, device=DEV), F, L)
    with pytest.raises(RuntimeError, match="sharpness_sigma"):
        ops.projection_loss(P, P, dists, idx, None, F, L, 0.0)
    with pytest.raises(RuntimeError, match="GPU tensors"):
        ops.projection_loss(P.cpu(), P, dists, idx, None, F, L, 0.75)


def test_empty_and_tiny_clouds():
    """P = 0 is a no-op; a cloud with fewer points than knn_k gets pytorch3d-style zero-padded lists (idx 0, distance 0)
    and the kernels still agree with the oracle entry by entry (NaN where the reference formula gives 0/0)."""
    F = torch.zeros(1, dtype=torch.int64, device=DEV)
    L0 = torch.zeros(1, dtype=torch.int64, device=DEV)
    empty = torch.zeros(0, 3, device=DEV)
    d0, i0 = ops.knn_points(empty, F, L0, 12)
    assert tuple(d0.shape) == (0, 12) and tuple(i0.shape) == (0, 12)
    assert tuple(ops.mollify_normals(empty, d0, i0, None, F, L0).shape) == (0, 3)
    l0, g0 = ops.projection_loss(empty, empty, d0, i0, None, F, L0, 0.75, want_grad=True)
    assert tuple(l0.shape) == (0,) and tuple(g0.shape) == (0, 3)
    l0, g0 = ops.repulsion_loss(empty, empty, i0, F, L0, 0.75, 2.0, want_grad=True)
    assert tuple(l0.shape) == (0, 3) and tuple(g0.shape) == (0, 3)

    rng = np.random.default_rng(8)
    pts = rng.normal(0, 1, (5, 3)).astype(np.float32)
    nrm = rng.normal(0, 1, (5, 3)).astype(np.float32)
    L = torch.full((1,), 5, dtype=torch.int64, device=DEV)
    dists, idx = ops.knn_points(_t(pts), F, L, 12)
    d_np, i_np = dists.cpu().numpy(), idx.cpu().numpy()
    assert (d_np[:, 5:] == 0).all() and (i_np[:, 5:] == 0).all() and (i_np[:, 0] == np.arange(5)).all()
    first_of = np.zeros(5, np.int64)
    moll = ops.mollify_normals(_t(nrm), dists, idx, None, F, L).cpu().numpy()
    assert np.allclose(moll, oracle.mollify_normals(nrm, d_np, i_np, None, first_of), rtol=1e-4, atol=1e-6, equal_nan=True)
    loss, grad = ops.projection_loss(_t(pts), _t(nrm), dists, idx, None, F, L, 0.75, want_grad=True)
    lo, go = oracle.projection_loss(pts, nrm, d_np, i_np, None, first_of, 0.75)
    assert np.allclose(loss.cpu().numpy(), lo, rtol=1e-3, atol=1e-7, equal_nan=True)
    assert np.allclose(grad.cpu().numpy(), go, rtol=1e-3, atol=1e-6, equal_nan=True)


def test_renderer_and_regulariser_share_one_neighbour_search(monkeypatch):
    """dss_amd.neighbours: once a regulariser exists, the renderer's kNN for the variance scale produces full lists and
    the regulariser of the same iteration reuses them; an in-place update of the points invalidates the cache; the
    variance scale is bit-identical to the dedicated K-th-distance kernel."""
    from dss_amd import neighbours
    from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    monkeypatch.setattr(neighbours, "_requested_k", 0)
    monkeypatch.setattr(neighbours, "_last", None)
    calls = {"lists": 0, "kth": 0}
    real_lists, real_kth = ops.knn_points, ops.knn_kth_sqdist

    def counting_lists(*a, **k):
        calls["lists"] += 1
        return real_lists(*a, **k)

    def counting_kth(*a, **k):
        calls["kth"] += 1
        return real_kth(*a, **k)

    monkeypatch.setattr(ops, "knn_points", counting_lists)
    monkeypatch.setattr(ops, "knn_kth_sqdist", counting_kth)
    pts, nrm = scenes.load_cloud("teapot")
    P = torch.nn.Parameter(_t(scenes.normalize_unit_sphere(pts))[None])          # (1,P,3) like Model.points
    normals = _t(nrm)[None]
    st = PointsRasterizationSettings(Vrk_invariant=True, image_size=64)
    raster = SurfaceSplatting(raster_settings=st)
    h_dedicated = raster._variance_scale(PointClouds3D(P, normals), st).clone()
    assert calls == {"lists": 0, "kth": 1}
    proj = ProjectionLoss(reduction="mean", filter_scale=2.0, knn_k=12)          # announces knn_k = 12
    assert neighbours.requested_k() == 12
    h_shared = raster._variance_scale(PointClouds3D(P, normals), st)
    assert calls == {"lists": 1, "kth": 1} and torch.equal(h_shared, h_dedicated)
    loss = proj(PointClouds3D(P, normals), rebuild_knn=True)                     # a different container, same points
    assert calls == {"lists": 1, "kth": 1}
    loss.backward()
    with torch.no_grad():
        P.add_(0.01 * torch.randn_like(P))                                       # the optimiser step
    proj(PointClouds3D(P, normals), rebuild_knn=True)
    assert calls == {"lists": 2, "kth": 1}
    rep = RepulsionLoss(reduction="mean", knn_k=8)                               # fewer neighbours: a slice of the lists
    rep(PointClouds3D(P, normals), rebuild_knn=True)
    assert calls == {"lists": 2, "kth": 1} and rep.knn_tree.idx.shape[1] == 8
    assert torch.equal(rep.knn_tree.idx, proj.knn_tree.idx[:, :8])
