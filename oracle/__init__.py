"""oracle -- TEST INFRASTRUCTURE ONLY.

numpy/ctypes front-end of ``oracle/dss_oracle.c``: the CPU restatement of the reference's EWA
splatting hot path, used as the parity checker for the HIP library.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product (``dss_amd``) never does.

``oracle.ref()`` returns the compiled *unmodified* reference CPU extension (``oracle/_ref``) when
it has been built (``make -C oracle ref``, needs /root/reference at build time only), else None.
"""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(with_ref: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when the reference checkout is present)."""
    targets = ["all"]
    if with_ref and os.path.isdir("/root/reference/DSS/csrc"):
        targets.append("ref")
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "dss_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def ref():
    """The compiled reference CPU extension (module ``dss_ref_cpu``) or None if not built."""
    d = os.path.join(_HERE, "_ref")
    if not os.path.isdir(d):
        return None
    for f in os.listdir(d):
        if f.startswith("dss_ref_cpu") and f.endswith(".so"):
            import torch  # noqa: F401  (the extension links against libtorch)
            spec = importlib.util.spec_from_file_location("dss_ref_cpu", os.path.join(d, f))
            mod = importlib.util.module_from_spec(spec)
            sys.modules.setdefault("dss_ref_cpu", mod)
            spec.loader.exec_module(mod)
            return mod
    return None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def splat_forward(points, ellipse, cutoff, radii, first_idx, num_pts, S, K, thr,
                  brute: bool = False, cpu_bbox_and: bool = False):
    """-> idx int32 (N,S,S,K), zbuf, qvalue f32 (N,S,S,K), occ f32 (N,S,S)."""
    points, ellipse, cutoff, radii = _f32(points), _f32(ellipse), _f32(cutoff), _f32(radii)
    first_idx, num_pts = _i64(first_idx), _i64(num_pts)
    N = first_idx.shape[0]
    idx = np.empty((N, S, S, K), np.int32)
    zbuf = np.empty((N, S, S, K), np.float32)
    qv = np.empty((N, S, S, K), np.float32)
    occ = np.empty((N, S, S), np.float32)
    L = _lib()
    if brute or cpu_bbox_and:
        rc = L.oracle_splat_forward_brute(_p(points), _p(ellipse), _p(cutoff), _p(radii), _p(first_idx),
                                          _p(num_pts), N, S, K, ctypes.c_float(thr),
                                          1 if cpu_bbox_and else 0, _p(idx), _p(zbuf), _p(qv), _p(occ))
    else:
        rc = L.oracle_splat_forward(_p(points), _p(ellipse), _p(cutoff), _p(radii), _p(first_idx),
                                    _p(num_pts), N, S, K, ctypes.c_float(thr),
                                    _p(idx), _p(zbuf), _p(qv), _p(occ))
    if rc != 0:
        raise RuntimeError("oracle_splat_forward failed: %d" % rc)
    return idx, zbuf, qv, occ


def visibility(idx, P):
    idx = np.ascontiguousarray(idx, np.int32)
    N, S, _, K = idx.shape
    vis = np.zeros((P,), np.uint8)
    _lib().oracle_visibility(_p(idx), N, S, K, ctypes.c_int64(P), _p(vis))
    return vis.astype(bool)


def backward_radius(radii, vis, first_idx, num_pts, radii_s):
    radii = _f32(radii)
    vis = np.ascontiguousarray(vis, np.uint8)
    first_idx, num_pts = _i64(first_idx), _i64(num_pts)
    N = first_idx.shape[0]
    rs = np.zeros((N,), np.float32)
    _lib().oracle_backward_radius(_p(radii), _p(vis), _p(first_idx), _p(num_pts), N,
                                  ctypes.c_float(radii_s), _p(rs))
    return rs


def occ_backward_fast(points, radii, vis, rs, grad_occ, first_idx, num_pts):
    points, radii, rs, grad_occ = _f32(points), _f32(radii), _f32(rs), _f32(grad_occ)
    vis = np.ascontiguousarray(vis, np.uint8)
    first_idx, num_pts = _i64(first_idx), _i64(num_pts)
    N, S = grad_occ.shape[0], grad_occ.shape[1]
    P = points.shape[0]
    g = np.zeros((P, 2), np.float32)
    _lib().oracle_occ_backward_fast(_p(points), _p(radii), _p(vis), _p(rs), _p(grad_occ), _p(first_idx),
                                    _p(num_pts), N, ctypes.c_int64(P), S, _p(g))
    return g


def occ_backward_slow_cpu(points, radii, grad_occ, first_idx, num_pts, radii_s):
    points, radii, grad_occ = _f32(points), _f32(radii), _f32(grad_occ)
    first_idx, num_pts = _i64(first_idx), _i64(num_pts)
    N, S = grad_occ.shape[0], grad_occ.shape[1]
    P = points.shape[0]
    g = np.zeros((P, 2), np.float32)
    _lib().oracle_occ_backward_slow_cpu(_p(points), _p(radii), _p(grad_occ), _p(first_idx), _p(num_pts),
                                        N, ctypes.c_int64(P), S, ctypes.c_float(radii_s), _p(g))
    return g


def occ_backward_slow_cuda(points, radii, grad_occ, first_idx, num_pts, radii_s):
    """RasterizePointsOccBackwardCudaKernel (rasterize_points.cu:672-757) restated -> (P,2)."""
    points, radii, grad_occ = _f32(points), _f32(radii), _f32(grad_occ)
    first_idx, num_pts = _i64(first_idx), _i64(num_pts)
    N, S = grad_occ.shape[0], grad_occ.shape[1]
    P = points.shape[0]
    g = np.zeros((P, 2), np.float32)
    _lib().oracle_occ_backward_slow_cuda(_p(points), _p(radii), _p(grad_occ), _p(first_idx), _p(num_pts),
                                         N, ctypes.c_int64(P), S, ctypes.c_float(radii_s), _p(g))
    return g


def zbuf_backward(idx, grad_zbuf, P, z_grad=None):
    idx = np.ascontiguousarray(idx, np.int32)
    grad_zbuf = _f32(grad_zbuf)
    N, S, _, K = idx.shape
    if z_grad is None:
        z_grad = np.zeros((P,), np.float32)
    _lib().oracle_zbuf_backward(_p(idx), _p(grad_zbuf), N, S, K, _p(z_grad))
    return z_grad


def clip_grad(grad, clip):
    g = _f32(grad).copy()
    _lib().oracle_clip_grad(_p(g), ctypes.c_int64(g.shape[0]), ctypes.c_float(clip))
    return g


def splat_backward(points, radii, idx, grad_occ, grad_zbuf, first_idx, num_pts, radii_s, clip=-1.0):
    """Full EllipticalRasterizer.backward (rasterizer.py:787-977 + clip hook :667-673):
    -> grad_pts (P,3), vis (P,) bool, rs (N,)."""
    P = np.asarray(points).shape[0]
    vis = visibility(idx, P)
    rs = backward_radius(radii, vis, first_idx, num_pts, radii_s)
    gxy = occ_backward_fast(points, radii, vis, rs, grad_occ, first_idx, num_pts)
    gz = np.zeros((P,), np.float32)
    if grad_zbuf is not None:
        zbuf_backward(idx, grad_zbuf, P, gz)
    g = np.concatenate([gxy, gz[:, None]], axis=1).astype(np.float32)
    if clip is not None and clip > 0:
        g = clip_grad(g, clip)
    return g, vis, rs


def blend_forward(idx, qv, occ, scaler, feat):
    idx = np.ascontiguousarray(idx, np.int32)
    qv, occ, scaler, feat = _f32(qv), _f32(occ), _f32(scaler), _f32(feat)
    N, S, _, K = idx.shape
    C = feat.shape[1]
    out = np.empty((N, S, S, C + 1), np.float32)
    _lib().oracle_blend_forward(_p(idx), _p(qv), _p(occ), _p(scaler), _p(feat), N, S, K, C, _p(out))
    return out


def blend_backward(grad_out, idx, qv, scaler, P):
    grad_out = _f32(grad_out)
    idx = np.ascontiguousarray(idx, np.int32)
    qv, scaler = _f32(qv), _f32(scaler)
    N, S, _, K = idx.shape
    C = grad_out.shape[-1] - 1
    gf = np.empty((P, C), np.float32)
    go = np.empty((N, S, S), np.float32)
    _lib().oracle_blend_backward(_p(grad_out), _p(idx), _p(qv), _p(scaler), N, S, K, C,
                                 ctypes.c_int64(P), _p(gf), _p(go))
    return gf, go


def point_setup(pts_world, normals, h, cloud_of, M, V, S, cutoff, sigma, vr6=None, frame_normals=None):
    """vr6 (P,6) + frame_normals (P,3): anisotropic source variance (rasterizer.py:256-291), see local_frames."""
    pts_world, normals, h, M, V = _f32(pts_world), _f32(normals), _f32(h), _f32(M), _f32(V)
    cloud_of = np.ascontiguousarray(cloud_of, np.int32)
    P = pts_world.shape[0]
    ps = np.empty((P, 3), np.float32)
    el = np.empty((P, 3), np.float32)
    ra = np.empty((P, 2), np.float32)
    sc = np.empty((P,), np.float32)
    cu = np.empty((P,), np.float32)
    if vr6 is not None:
        vr6, frame_normals = _f32(vr6), _f32(frame_normals)
    _lib().oracle_point_setup(_p(pts_world), _p(normals), _p(h), _p(cloud_of), _p(M), _p(V),
                              ctypes.c_int64(P), S, ctypes.c_float(cutoff), ctypes.c_float(sigma),
                              _p(vr6) if vr6 is not None else None,
                              _p(frame_normals) if vr6 is not None else None,
                              _p(ps), _p(el), _p(ra), _p(sc), _p(cu))
    return ps, el, ra, sc, cu


def local_frames(points, knn_idx_packed):
    """PCA frames of the K-neighbourhoods (mathHelper.py:34-92) -> (vr6 (P,6), frame normal (P,3), curvature (P,3)
    ascending).  knn_idx_packed (P,K) int64 holds PACKED point ids (self included)."""
    points = _f32(points)
    idx = np.ascontiguousarray(knn_idx_packed, np.int64)
    P, K = idx.shape
    vr6 = np.empty((P, 6), np.float32)
    fn = np.empty((P, 3), np.float32)
    cv = np.empty((P, 3), np.float32)
    _lib().oracle_local_frames(_p(points), _p(idx), ctypes.c_int64(P), K, _p(vr6), _p(fn), _p(cv))
    return vr6, fn, cv


def phong_forward(points, normals, rgb, cloud_of, ambient, diffuse_color, specular_color, light_vec, point_lights,
                  cam_center, shininess):
    """LightingTexture.forward (texture.py:65-125, lighting.py:10-172) -> (shaded, diffuse, specular), each (P,3)."""
    points, normals, rgb = _f32(points), _f32(normals), _f32(rgb)
    ambient, kd, ks, lv, cam = _f32(ambient), _f32(diffuse_color), _f32(specular_color), _f32(light_vec), _f32(cam_center)
    cloud_of = np.ascontiguousarray(cloud_of, np.int32)
    P, L = points.shape[0], kd.shape[1]
    out, dif, spc = (np.empty((P, 3), np.float32) for _ in range(3))
    _lib().oracle_phong_forward(_p(points), _p(normals), _p(rgb), _p(cloud_of), ctypes.c_int64(P), _p(ambient), _p(kd),
                                _p(ks), _p(lv), L, int(bool(point_lights)), _p(cam), ctypes.c_float(shininess), _p(out),
                                _p(dif), _p(spc))
    return out, dif, spc


def _u8_or_null(a):
    """(array kept alive, pointer) of an optional uint8 mask."""
    if a is None:
        return None, None
    x = np.ascontiguousarray(a, np.uint8)
    return x, _p(x)


def mollify_normals(normals, knn_d2, knn_idx, keep, first_of):
    """SurfaceLoss._denoise_normals with get_phi weights (losses.py:181-222, 262-278), packed -> (P,3)."""
    normals, knn_d2 = _f32(normals), _f32(knn_d2)
    knn_idx, first_of = np.ascontiguousarray(knn_idx, np.int64), np.ascontiguousarray(first_of, np.int64)
    P, K = knn_idx.shape
    out = np.empty((P, 3), np.float32)
    kp = _u8_or_null(keep)
    _lib().oracle_mollify_normals(_p(normals), _p(knn_d2), _p(knn_idx), kp[1], _p(first_of), ctypes.c_int64(P), K, _p(out))
    return out


def projection_loss(points, mollified, knn_d2, knn_idx, visible, first_of, sigma, grad_loss=None):
    """ProjectionLoss.compute (losses.py:296-392) per point -> (loss (P,), grad_points (P,3))."""
    points, mollified, knn_d2 = _f32(points), _f32(mollified), _f32(knn_d2)
    knn_idx, first_of = np.ascontiguousarray(knn_idx, np.int64), np.ascontiguousarray(first_of, np.int64)
    P, K = knn_idx.shape
    loss, grad = np.empty((P,), np.float32), np.empty((P, 3), np.float32)
    vp = _u8_or_null(visible)
    gl = None if grad_loss is None else _f32(grad_loss)
    _lib().oracle_projection_loss(_p(points), _p(mollified), _p(knn_d2), _p(knn_idx), vp[1], _p(first_of),
                                  ctypes.c_int64(P), K, ctypes.c_float(sigma), None if gl is None else _p(gl), _p(loss),
                                  _p(grad))
    return loss, grad


def repulsion_loss(points, mollified, knn_idx, first_of, inv_sigma_of, sigma, grad_loss=None):
    """RepulsionLoss.compute (losses.py:395-492) per point -> (loss (P,3), grad_points (P,3))."""
    points, mollified, inv_sigma_of = _f32(points), _f32(mollified), _f32(inv_sigma_of)
    knn_idx, first_of = np.ascontiguousarray(knn_idx, np.int64), np.ascontiguousarray(first_of, np.int64)
    P, K = knn_idx.shape
    loss, grad = np.empty((P, 3), np.float32), np.empty((P, 3), np.float32)
    gl = None if grad_loss is None else _f32(grad_loss)
    _lib().oracle_repulsion_loss(_p(points), _p(mollified), _p(knn_idx), _p(first_of), _p(inv_sigma_of),
                                 ctypes.c_int64(P), K, ctypes.c_float(sigma), None if gl is None else _p(gl), _p(loss),
                                 _p(grad))
    return loss, grad


def image_loss(rgba, img, mask, lambda_rgb, lambda_sil, want_grad=True):
    """Trainer.calc_dr_loss (trainer.py:332-372) and its gradient w.r.t. the rendered RGBA image ->
    (losses (4,) = total, weighted rgb, weighted silhouette, IoU term; grad_rgba (N,H,W,4) or None)."""
    rgba, img, mask = _f32(rgba), _f32(img), _f32(mask)
    N, H, W = mask.shape
    losses = np.empty((4,), np.float32)
    grad = np.empty((N, H, W, 4), np.float32) if want_grad else None
    _lib().oracle_image_loss(_p(rgba), _p(img), _p(mask), N, H, W, ctypes.c_float(lambda_rgb), ctypes.c_float(lambda_sil),
                             _p(losses), None if grad is None else _p(grad))
    return losses, grad


def grid_sample_points(mask, grid):
    """F.grid_sample(mask[:, None], grid[:, None], 'bilinear', 'reflection', align_corners=False) restated ->
    (N,P); mask (N,H,W), grid (N,P,2) in [-1,1] (x, y)."""
    mask, grid = _f32(mask), _f32(grid)
    N, H, W = mask.shape
    P = grid.shape[1]
    out = np.empty((N, P), np.float32)
    _lib().oracle_grid_sample_points(_p(mask), _p(grid), N, ctypes.c_int64(P), H, W, _p(out))
    return out


def points_inmask(points, M, mask, visible=None):
    """In-mask filter of point_modeling.py:183-208 -> uint8 (P,)."""
    points, M, mask = _f32(points), _f32(M), _f32(mask)
    N, H, W = mask.shape
    P = points.shape[0]
    out = np.empty((P,), np.uint8)
    vp = _u8_or_null(visible)
    _lib().oracle_points_inmask(_p(points), _p(M), _p(mask), vp[1], N, ctypes.c_int64(P), H, W, _p(out))
    return out
