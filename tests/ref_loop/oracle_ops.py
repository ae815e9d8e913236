"""TEST DOUBLE (CPU): the handful of `dss_amd.ops` entry points the drop-in renderer calls (the fused pair included: it is
the renderer's default path), answered by the oracle.

Only `tests/` may use the oracle, and only as a checker -- here it stands in for `libdss_hip.so` on the GPU-less build
container so that the reference's own `train_mvr.py` can be driven through the drop-in classes end to end
(tests/test_reference_loop_cpu.py).  On a GPU the launcher does not install it: the real HIP ops run.  Never imported by
the product."""
import numpy as np
import torch

import oracle


def _np(t):
    return t.detach().cpu().numpy()


def _cloud_of(first, num, P):
    cloud_of = np.full((P,), -1, np.int32)
    for n, (f, c) in enumerate(zip(_np(first).tolist(), _np(num).tolist())):
        cloud_of[f:f + c] = n
    return cloud_of


def point_setup(world, normals, h, M, V, znear, zfar, first, num, image_size, cutoff_threshold, antialiasing_sigma=1.0,
                backface_culling=False, shared_cloud=False, vr6=None, frame_normals=None):
    N, Pw = first.shape[0], world.shape[0]
    if shared_cloud:
        world_p, normals_p = world.repeat(N, 1), normals.repeat(N, 1)
        h_p = h.repeat(N) if h.numel() == Pw and not (h.numel() == N and Pw == N) else h
    else:
        world_p, normals_p, h_p = world, normals, h
    P = world_p.shape[0]
    cloud_of = _cloud_of(first, num, P)
    if h_p.numel() != P:  # per cloud -> per point
        h_p = h_p[torch.from_numpy(np.maximum(cloud_of, 0)).long()]
    ps, el, ra, sc, cu = oracle.point_setup(_np(world_p), _np(normals_p), _np(h_p), cloud_of, _np(M), _np(V), int(image_size),
                                            float(cutoff_threshold), float(antialiasing_sigma))
    zview = ps[:, 2]
    c = np.maximum(cloud_of, 0)
    valid = (cloud_of >= 0) & (zview >= _np(znear)[c]) & (zview <= _np(zfar)[c])
    if backface_culling:
        nz = (_np(normals_p) * _np(V)[c][:, :3, 2]).sum(1)
        valid &= nz < 0
    ps[~valid] = (0.0, 0.0, -1.0)
    ra[~valid] = 0.0
    t = torch.from_numpy
    return dict(pts_screen=t(ps), ellipse_params=t(el), radii=t(ra), scaler=t(sc * valid), cutoff_threshold=t(cu),
                valid=t(valid))


def project_backward(world, M, V, first, num, grad_screen, valid, shared_cloud=False, clip=-1.0):
    N, Pw = first.shape[0], world.shape[0]
    g = grad_screen.clone()
    if clip is not None and clip > 0:
        nrm = g.norm(dim=1, keepdim=True)
        g = torch.where(nrm > clip, g * (clip / nrm.clamp(min=1e-30)), g)
    g = g * valid.to(g.dtype)[:, None]
    with torch.enable_grad():
        w = world.detach().double().requires_grad_(True)
        ones = torch.ones(Pw, 1, dtype=torch.float64)
        hom = torch.cat([w, ones], 1)
        total = 0.0
        for n in range(N):
            f, c = int(first[n]), int(num[n])
            rows = hom if shared_cloud else hom[f:f + c]
            clipc = rows @ M[n].double()
            zv = (rows @ V[n].double())[:, 2]
            scr = torch.stack([clipc[:, 0] / clipc[:, 3], clipc[:, 1] / clipc[:, 3], zv], 1)
            total = total + (scr * g[f:f + c].double()).sum()
        (gw,) = torch.autograd.grad(total, w)
    return gw.float()


def splat_points(points, ellipse, cutoff, radii, first, num, thr, S, K, bin_size=None, max_points_per_bin=None,
                 return_visible=False):
    idx, zbuf, qv, occ = oracle.splat_forward(_np(points), _np(ellipse), _np(cutoff), _np(radii), _np(first), _np(num),
                                              int(S), int(K), float(thr))
    out = [torch.from_numpy(np.ascontiguousarray(a)) for a in (idx, zbuf, qv, occ)]
    if return_visible:
        out.append(torch.from_numpy(oracle.visibility(idx, points.shape[0]).astype(bool)))
    return tuple(out)


def splat_backward(points, radii, visible, idx, grad_occ, grad_zbuf, first, num, radii_s, clip=-1.0, return_rs=False):
    N, S = idx.shape[0], idx.shape[1]
    go = np.zeros((N, S, S), np.float32) if grad_occ is None else np.ascontiguousarray(_np(grad_occ), np.float32)
    gz = None if grad_zbuf is None else _np(grad_zbuf)
    g, _vis, rs = oracle.splat_backward(_np(points), _np(radii), _np(idx), go, gz, _np(first), _np(num), float(radii_s),
                                        float(clip))
    g = torch.from_numpy(g)
    return (g, torch.from_numpy(rs)) if return_rs else g


def blend_forward(idx, qvalue, occupancy, scaler, features, return_wsum=False):
    out = torch.from_numpy(oracle.blend_forward(_np(idx), _np(qvalue), _np(occupancy), _np(scaler), _np(features)))
    if not return_wsum:
        return out
    w = torch.exp(-0.5 * qvalue) * scaler[idx.clamp(min=0).long()] * (idx >= 0)
    return out, w.sum(-1).clamp(min=1e-4)


def blend_backward(grad_out, idx, qvalue, scaler, num_points, geometry=None, wsum=None, image_size=None, rows=None):
    gf, go = oracle.blend_backward(_np(grad_out), _np(idx), _np(qvalue), _np(scaler), int(num_points))
    return torch.from_numpy(gf), torch.from_numpy(go)


def knn_kth_sqdist(points, first, num, K=7, radius=None):
    out = torch.zeros(points.shape[0])
    for f, c in zip(first.tolist(), num.tolist()):
        if c == 0:
            continue
        p = points[f:f + c].double()
        d2 = torch.cdist(p, p) ** 2
        if radius is not None and radius > 0:   # frnn_grid_points(K, r): the farthest of the K - 1 neighbours found within r, else -1
            near = torch.topk(d2, min(K, c), dim=1, largest=False).values[:, 1:]
            near = torch.where(near <= float(radius) ** 2, near, torch.full_like(near, -1.0))
            out[f:f + c] = (near.amax(dim=1) if near.shape[1] else torch.full((c,), -1.0, dtype=torch.float64)).float()
        else:
            out[f:f + c] = torch.kthvalue(d2, min(K, c), dim=1).values.float()  # self included (distance 0) as the first
    return out


def cloud_mean_clamp(values, first, num, scale, lo, hi, fallback, min_points):
    out = []
    for f, c in zip(first.tolist(), num.tolist()):
        out.append(float(fallback) if c < min_points else float(min(max(float(values[f:f + c].double().mean()) * scale, lo), hi)))
    return torch.tensor(out, dtype=torch.float32)


def _kept(world, Vn, zn, zf):
    """the depth test of the setup, same fp32 expression"""
    w, v = world.float(), Vn.float()
    z = w[:, 0] * v[0, 2] + w[:, 1] * v[1, 2] + w[:, 2] * v[2, 2] + 1.0 * v[3, 2]
    return (z >= zn) & (z <= zf)


def renderable_mean_clamp(values, world, V, znear, zfar, first, num, shared_cloud, scale, lo, hi, fallback, min_points):
    """ops.renderable_mean_clamp: per camera the sum over the points it keeps divided by the LARGEST kept count (the
    reference's mean over the padded batch, rasterizer.py:320-326); values (Pw,) or (N,Pw)"""
    N = V.shape[0]
    sums, cnts = [], []
    for n in range(N):
        f, c = (0, int(num[0])) if shared_cloud else (int(first[n]), int(num[n]))
        ok = _kept(world[f:f + c], V[n], znear[n], zfar[n])
        vals = values[n, f:f + c] if values.dim() == 2 else values[f:f + c]
        sums.append(float((vals[ok].float() * scale).double().sum()))
        cnts.append(int(ok.sum()))
    pmax = max(cnts)
    return torch.tensor([float(min(max(s / pmax, lo), hi)) if (c >= min_points and pmax > 0) else float(min(max(fallback, lo), hi))
                         for s, c in zip(sums, cnts)], dtype=torch.float32)


def knn_kth_sqdist_view(points, first, num, K, V, znear, zfar, shared_cloud, radius=None):
    """ops.knn_kth_sqdist_view: the search among the points the camera keeps"""
    n_cams, P = V.shape[0], points.shape[0]
    out = torch.zeros((n_cams, P) if shared_cloud else (P,))
    one = torch.zeros(1, dtype=torch.int64)
    for n in range(n_cams):
        f, c = (0, int(num[0])) if shared_cloud else (int(first[n]), int(num[n]))
        ok = _kept(points[f:f + c], V[n], znear[n], zfar[n])
        sub = points[f:f + c][ok]
        vals = knn_kth_sqdist(sub, one, torch.tensor([sub.shape[0]]), K, radius) if sub.shape[0] else torch.zeros(0)
        dst = out[n, f:f + c] if shared_cloud else out[f:f + c]
        dst[ok] = vals
    return out


def _splat_points_occ_fast_cuda_backward(points_sorted, radii_sorted, rs, grad_occ, num_points_per_cloud,
                                         cloud_to_packed_first_idx, points_grid_off=None, grid_params=None):
    """ext.cpp:14 -- the reference calls it with the VISIBLE points only; the FRNN grid arguments are not needed"""
    P = points_sorted.shape[0]
    pts = _np(points_sorted)
    if pts.shape[1] == 2:
        pts = np.concatenate([pts, np.zeros((P, 1), np.float32)], 1)
    g = oracle.occ_backward_fast(pts, _np(radii_sorted), np.ones(P, bool), _np(rs), _np(grad_occ),
                                 _np(cloud_to_packed_first_idx), _np(num_points_per_cloud))
    return torch.from_numpy(np.ascontiguousarray(g[:, :2]))


def _backward_zbuf(idx, grad_zbuf, point_z_grad):
    """ext.cpp:17: in place on (P,1)"""
    P = point_z_grad.shape[0]
    gz = np.zeros((P,), np.float32)
    oracle.zbuf_backward(_np(idx), _np(grad_zbuf), P, gz)
    point_z_grad += torch.from_numpy(gz)[:, None]


def render_forward(world, normals, h, M, V, znear, zfar, first, num, features, image_size, points_per_pixel,
                   cutoff_threshold, depth_merging_thres, antialiasing_sigma=1.0, backface_culling=False, shared_cloud=False,
                   rows=None, out_image=None, out_visible=None, vr6=None, frame_normals=None, want_zbuf=True,
                   workspace_state=1, order_refresh=0):
    """the fused forward (dss_render_forward) = setup -> rasterizer -> blend, composed from the doubles above
    (`order_refresh`: the cached point order of the HIP binning has no counterpart here -- the outputs do not depend on it)"""
    assert rows is None and out_image is None and vr6 is None
    o = point_setup(world, normals, h, M, V, znear, zfar, first, num, image_size, cutoff_threshold, antialiasing_sigma,
                    backface_culling, shared_cloud)
    idx, zbuf, qv, occ, vis = splat_points(o["pts_screen"], o["ellipse_params"], o["cutoff_threshold"], o["radii"], first, num,
                                           depth_merging_thres, image_size, points_per_pixel, return_visible=True)
    image, wsum = blend_forward(idx, qv, occ, o["scaler"], features, return_wsum=True)
    o.update(idx=idx, zbuf=zbuf, qvalue=qv, occupancy=occ, visible=vis, image=image, wsum=wsum)
    return o


def render_backward(grad_out, idx, qvalue, wsum, scaler, points, radii, visible, first, num, radii_s, clip=-1.0,
                    with_features=True, return_rs=False, image_size=None, rows=None, out=None, gather_only_rs=None,
                    project=None):
    """the fused backward (dss_render_backward): blend backward + occupancy surrogate (+ clip, + projection backward)"""
    assert rows is None and out is None and gather_only_rs is None
    gf, gocc = blend_backward(grad_out, idx, qvalue, scaler, points.shape[0])
    g = splat_backward(points, radii, visible, idx, gocc, None, first, num, radii_s, clip)
    if project is not None:
        world, M = project
        ident = torch.eye(4).expand(M.shape[0], 4, 4).contiguous()   # (the z gradient is 0 on this path: V is not needed)
        g = project_backward(world, M, ident, first, num, g, torch.ones(points.shape[0], dtype=torch.bool), False)
    return (gf if with_features else None, g)


def install(ops_module) -> None:
    for name in ("point_setup", "project_backward", "splat_points", "splat_backward", "blend_forward", "blend_backward",
                 "knn_kth_sqdist", "cloud_mean_clamp", "renderable_mean_clamp", "knn_kth_sqdist_view", "_splat_points_occ_fast_cuda_backward", "_backward_zbuf",
                 "render_forward", "render_backward"):
        setattr(ops_module, name, globals()[name])
