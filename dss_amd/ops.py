"""Tensor-level operators over the C ABI: ALL seven functions the reference binds in ``DSS._C``
(DSS/csrc/ext.cpp:5-18: ``splat_points``, ``_splat_points_naive``, ``_splat_points_occ_backward``,
``_rasterize_coarse``, ``_rasterize_fine``, ``_splat_points_occ_fast_cuda_backward``, ``_backward_zbuf``), same names and
argument meaning, plus the fused entry points.

All inputs must be GPU tensors; outputs are freshly allocated on the input's device
(rasterize_points.cu:632-637) except ``_backward_zbuf`` which accumulates in place
(rasterize_points.h:388-392).
"""
from typing import Optional, Tuple

import weakref

import torch

from . import _lib

_f32, _i32, _i64, _u8 = torch.float32, torch.int32, torch.int64, torch.uint8


def _as_u8(mask):
    """bool -> uint8 without a copy kernel (same storage, values 0/1)."""
    return mask.view(_u8) if mask.dtype == torch.bool else mask


def _check_raster_inputs(points, ellipse_params, cutoff_thres, radii, first_idx, num_pts):
    # shape checks of RasterizePoints, rasterize_points.h:474-488
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("points must have shape (P, 3), got %s" % (tuple(points.shape),))
    P = points.shape[0]
    if tuple(radii.shape) != (P, 2):
        raise RuntimeError("radii must have shape (%d, 2), got %s" % (P, tuple(radii.shape)))
    if tuple(ellipse_params.shape) != (P, 3):
        raise RuntimeError("ellipse_params must have shape (%d, 3), got %s" % (P, tuple(ellipse_params.shape)))
    if tuple(cutoff_thres.shape) != (P,):
        raise RuntimeError("cutoff_thres must have shape (%d,), got %s" % (P, tuple(cutoff_thres.shape)))
    if first_idx.shape != num_pts.shape or first_idx.dim() != 1:
        raise RuntimeError("cloud_to_packed_first_idx and num_points_per_cloud must both be (N,)")
    return P


def splat_points(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                 num_points_per_cloud, depth_merging_thres: float, image_size: int,
                 points_per_pixel: int, bin_size: Optional[int] = None,
                 max_points_per_bin: Optional[int] = None, *, rows: Optional[Tuple[int, int]] = None,
                 return_visible: bool = False):
    """``DSS._C.splat_points`` (ext.cpp:8, rasterize_points.h:461-525).

    Returns ``(idx int32 (N,S,S,K), zbuf, qvalue f32 (N,S,S,K), occupancy f32 (N,S,S))``.
    ``bin_size == 0`` scans every cloud per tile (the reference's naive mode); any other value
    (or None) builds compacted screen-tile lists.  ``max_points_per_bin`` is accepted for
    signature compatibility and ignored: tile lists are compacted, never truncated.
    ``rows=(row0,row1)`` renders only that row band (outputs are band shaped);
    ``return_visible`` appends the per-point visibility mask (bool (P,)).
    """
    lib = _lib.load()
    P = _check_raster_inputs(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                             num_points_per_cloud)
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    ellipse_params = _lib.require_gpu(ellipse_params, "ellipse_params", _f32)
    cutoff_thres = _lib.require_gpu(cutoff_thres, "cutoff_thres", _f32)
    radii = _lib.require_gpu(radii, "radii", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, S, K = first.shape[0], int(image_size), int(points_per_pixel)
    row0, row1 = (0, S) if rows is None else (int(rows[0]), int(rows[1]))
    nrows = max(row1 - row0, 0)
    bs = 1 if bin_size is None else int(bin_size)
    with torch.cuda.device(dev):
        idx = torch.empty((N, nrows, S, K), dtype=_i32, device=dev)
        zbuf = torch.empty((N, nrows, S, K), dtype=_f32, device=dev)
        qv = torch.empty((N, nrows, S, K), dtype=_f32, device=dev)
        occ = torch.empty((N, nrows, S), dtype=_f32, device=dev)
        vis = torch.empty((P,), dtype=_u8, device=dev) if return_visible else None
        if nrows == 0:  # a rank whose row band is empty (RowPartition with S < world_size * band): nothing to launch
            if vis is not None:
                vis.zero_()
                return idx, zbuf, qv, occ, vis.view(torch.bool)
            return idx, zbuf, qv, occ
        nbytes = lib.dss_splat_forward_workspace(N, P, S, K, bs)
        ws = _lib.workspace(dev, nbytes)
        rc = lib.dss_splat_forward(_lib.ptr(points), _lib.ptr(ellipse_params), _lib.ptr(cutoff_thres),
                                   _lib.ptr(radii), _lib.ptr(first), _lib.ptr(num), N, P,
                                   float(depth_merging_thres), S, K, bs, row0, row1,
                                   _lib.ptr(idx), _lib.ptr(zbuf), _lib.ptr(qv), _lib.ptr(occ), _lib.ptr(vis),
                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_splat_forward")
    if return_visible:
        return idx, zbuf, qv, occ, vis.view(torch.bool)  # zero-copy reinterpretation (values are 0/1)
    return idx, zbuf, qv, occ


def _splat_points_naive(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                        num_points_per_cloud, depth_merging_thres, image_size, points_per_pixel):
    """``DSS._C._splat_points_naive`` (ext.cpp:9)."""
    return splat_points(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                        num_points_per_cloud, depth_merging_thres, image_size, points_per_pixel, 0, 0)


def _rasterize_coarse(points, radii, cloud_to_packed_first_idx, num_points_per_cloud, image_size: int, bin_size: int,
                      max_points_per_bin: int):
    """``DSS._C._rasterize_coarse`` (ext.cpp:11, rasterize_points.h:167-203): the binning pass alone.  The reference
    returns a dense ``(N, B, B, M)`` int32 table; here ``bin_points`` is an OPAQUE uint8 tensor (the tile-list workspace
    of ``dss_splat_bin``: per-tile sub-list counters + fixed-capacity id lists over 8x8-pixel tiles) that is only meant
    to be handed to :func:`_rasterize_fine`.  ``bin_size`` / ``max_points_per_bin`` are accepted and ignored (tile size
    and list capacity are chosen by the library; lists never truncate)."""
    lib = _lib.load()
    src = _bin_source(points, radii)   # what the CALLER handed over (before any normalisation copy)
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    radii = _lib.require_gpu(radii, "radii", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    if points.dim() != 2 or points.shape[1] != 3 or tuple(radii.shape) != (points.shape[0], 2):
        raise RuntimeError("points must be (P,3) and radii (P,2)")
    N, P, S = first.shape[0], points.shape[0], int(image_size)
    with torch.cuda.device(dev):
        nbytes = lib.dss_splat_forward_workspace(N, P, S, 1, 1)
        bin_points = torch.empty(nbytes, dtype=_u8, device=dev)
        if P > 0:
            rc = lib.dss_splat_bin(_lib.ptr(points), _lib.ptr(radii), _lib.ptr(first), _lib.ptr(num), N, P, S, 0, S,
                                   _lib.ptr(bin_points), nbytes, _lib.stream_ptr(dev))
            _lib.check(rc, "dss_splat_bin")
    # what the fine pass needs besides the lists, and what the lists were built from.  Kept in a registry keyed by the
    # buffer's address (not as a Python attribute of the tensor, which views, autograd saves and `detach()` drop): any
    # tensor that still refers to this storage finds it; a copy (clone / to) does not and is refused with a clear message.
    # keyed by the STORAGE address, so that a view with a storage offset still finds its entry
    _bin_registry[bin_points.untyped_storage().data_ptr()] = (weakref.ref(bin_points), first, num, N, P, S, src)
    if len(_bin_registry) > 64:
        for k in [k for k, v in _bin_registry.items() if v[0]() is None]:
            del _bin_registry[k]
    return bin_points


_bin_registry = {}


def _bin_source(points, radii):
    """identity of the caller's tensors as handed over: address, layout and version counter (a non-contiguous input is
    normalised into a temporary inside the call; comparing the temporary's address would reject the same tensor next time,
    or accept another one that recycled the address)"""
    return tuple((t.data_ptr(), tuple(t.shape), tuple(t.stride()), t._version) for t in (points, radii))


def _rasterize_fine(points, ellipse_params, cutoff_thres, radii, bin_points, depth_merging_thres: float, image_size: int,
                    bin_size: int, points_per_pixel: int):
    """``DSS._C._rasterize_fine`` (ext.cpp:12, rasterize_points.h:257-285) on the ``bin_points`` of
    :func:`_rasterize_coarse` -> ``(idx, zbuf, qvalue, occupancy)`` like ``splat_points``."""
    lib = _lib.load()
    meta = _bin_registry.get(bin_points.untyped_storage().data_ptr()) \
        if isinstance(bin_points, torch.Tensor) and bin_points.is_cuda else None
    if meta is None or meta[0]() is None:
        raise RuntimeError("bin_points must be the tensor dss_amd.ops._rasterize_coarse returned, or a view of it (opaque "
                           "tile lists, not the reference's dense (N,B,B,M) table; a clone / device copy is not accepted)")
    whole, first, num, N, P, S, src = meta
    whole = whole()   # the tensor _rasterize_coarse returned (bin_points may be a view of it)
    if int(image_size) != S or points.shape[0] != P:
        raise RuntimeError("bin_points were built for image_size=%d and %d points" % (S, P))
    if src != _bin_source(points, radii):
        raise RuntimeError("bin_points were built from other points / radii tensors (or these were modified since)")
    _check_raster_inputs(points, ellipse_params, cutoff_thres, radii, first, num)
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    ellipse_params = _lib.require_gpu(ellipse_params, "ellipse_params", _f32)
    cutoff_thres = _lib.require_gpu(cutoff_thres, "cutoff_thres", _f32)
    radii = _lib.require_gpu(radii, "radii", _f32)
    K = int(points_per_pixel)
    with torch.cuda.device(dev):
        idx = torch.empty((N, S, S, K), dtype=_i32, device=dev)
        zbuf = torch.empty((N, S, S, K), dtype=_f32, device=dev)
        qv = torch.empty((N, S, S, K), dtype=_f32, device=dev)
        occ = torch.empty((N, S, S), dtype=_f32, device=dev)
        rc = lib.dss_splat_fine(_lib.ptr(points), _lib.ptr(ellipse_params), _lib.ptr(cutoff_thres), _lib.ptr(radii),
                                _lib.ptr(first), _lib.ptr(num), N, P, float(depth_merging_thres), S, K, 0, S, _lib.ptr(idx),
                                _lib.ptr(zbuf), _lib.ptr(qv), _lib.ptr(occ), None, _lib.ptr(whole) if P > 0 else None,
                                whole.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_splat_fine")
    return idx, zbuf, qv, occ


def _splat_points_occ_backward(points, radii, grad_occ, cloud_to_packed_first_idx, num_points_per_cloud, radii_s: float,
                               depth_merging_thres: float = 0.0):
    """``DSS._C._splat_points_occ_backward`` on GPU tensors (ext.cpp:10, 16; rasterize_points.cu:672-822): the
    box-supported occupancy surrogate over ALL points handed in -> (P,2).  Not on the training path
    (``backward_occ_fast = True``, rasterizer.py:816).  ``depth_merging_thres`` is unused, as in the reference."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    radii = _lib.require_gpu(radii, "radii", _f32)
    grad_occ = _lib.require_gpu(grad_occ, "grad_occ", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    if points.dim() != 2 or points.shape[1] != 3 or tuple(radii.shape) != (points.shape[0], 2) or first.shape != num.shape:
        raise RuntimeError("points must be (P,3), radii (P,2), first_idx / num_points (N,)")  # rasterize_points.h:357-361
    N, P = first.shape[0], points.shape[0]
    if grad_occ.dim() != 3 or grad_occ.shape[0] != N or grad_occ.shape[1] != grad_occ.shape[2]:
        raise RuntimeError("grad_occ must be (N,S,S) with N=%d, got %s" % (N, tuple(grad_occ.shape)))
    with torch.cuda.device(dev):
        grad = torch.empty((P, 2), dtype=_f32, device=dev)
        rc = lib.dss_occ_backward_box(_lib.ptr(points), _lib.ptr(radii), _lib.ptr(grad_occ), _lib.ptr(first), _lib.ptr(num),
                                      N, P, grad_occ.shape[1], float(radii_s), _lib.ptr(grad), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_occ_backward_box")
    return grad


def _splat_points_occ_fast_cuda_backward(points_sorted, radii_sorted, rs, grad_occ, num_points_per_cloud,
                                         cloud_to_packed_first_idx, points_grid_off=None, grid_params=None):
    """``DSS._C._splat_points_occ_fast_cuda_backward`` (ext.cpp:14, rasterize_points_backward.cu:227-322) -> (P,2) in the
    order of ``points_sorted``.  Every point handed in takes part (the reference passes the visible points only,
    rasterizer.py:863-864, 951-952).  ``points_grid_off`` / ``grid_params`` -- the FRNN grid that only accelerates the
    reference's pixel-centric search -- are accepted and ignored: the gather kernel needs no grid, so the points need
    not be sorted either (and the last-cell bug of :124-126 cannot occur)."""
    all_points = torch.ones(points_sorted.shape[0], dtype=_u8, device=points_sorted.device)
    return occ_backward(points_sorted, radii_sorted, all_points, rs, grad_occ, cloud_to_packed_first_idx,
                        num_points_per_cloud)[:, :2].contiguous()


def backward_radius(radii, visible, cloud_to_packed_first_idx, num_points_per_cloud, radii_s: float):
    """Search radius of the backward pass, rasterizer.py:885-888 -> f32 (N,)."""
    lib = _lib.load()
    radii = _lib.require_gpu(radii, "radii", _f32)
    dev = radii.device
    vis = _lib.require_gpu(_as_u8(visible), "visible", _u8)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, P = first.shape[0], radii.shape[0]
    with torch.cuda.device(dev):
        rs = torch.empty((N,), dtype=_f32, device=dev)
        ws = _lib.workspace(dev, lib.dss_backward_radius_workspace(N, P))
        rc = lib.dss_backward_radius(_lib.ptr(radii), _lib.ptr(vis), _lib.ptr(first), _lib.ptr(num), N, P,
                                     float(radii_s), _lib.ptr(rs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_backward_radius")
    return rs


def _pixel_strided(t, N, rows, S):
    """(tensor, pixel stride) for an (N,rows,S) float32 GPU view that is dense up to a constant element
    stride per pixel (e.g. ``grad_image[..., 3]``); anything else is made contiguous."""
    if t.is_cuda and t.dtype == _f32 and tuple(t.shape) == (N, rows, S) and t.numel() > 0:
        c = t.stride(2)
        if c >= 1 and t.stride(1) == S * c and t.stride(0) == rows * S * c:
            return t, c
    return _lib.require_gpu(t, "grad_occ", _f32), 1


def occ_backward(points, radii, visible, rs, grad_occ, cloud_to_packed_first_idx, num_points_per_cloud,
                 image_size: Optional[int] = None, rows: Optional[Tuple[int, int]] = None, clip: float = -1.0):
    """Occupancy surrogate gradient -> (P,3) with z column 0.  Replaces the FRNN grid build
    (rasterizer.py:889-950) + ``DSS._C._splat_points_occ_fast_cuda_backward`` (ext.cpp:14).
    ``grad_occ`` may be a strided channel view of an image gradient (read in place).  ``clip > 0`` fuses
    the per-point clip hook (valid only when no zbuf gradient / cross-rank reduction follows)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    radii = _lib.require_gpu(radii, "radii", _f32)
    vis = _lib.require_gpu(_as_u8(visible), "visible", _u8)
    rs = _lib.require_gpu(rs, "rs", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, P = first.shape[0], points.shape[0]
    S = int(image_size) if image_size is not None else grad_occ.shape[2]
    row0, row1 = (0, S) if rows is None else (int(rows[0]), int(rows[1]))
    if tuple(grad_occ.shape) != (N, row1 - row0, S):
        raise RuntimeError("grad_occ must have shape (%d, %d, %d), got %s" % (N, row1 - row0, S, tuple(grad_occ.shape)))
    if row1 <= row0:  # empty band: no pixel contributes
        return torch.zeros((P, 3), dtype=_f32, device=dev)
    grad_occ, gstride = _pixel_strided(grad_occ, N, row1 - row0, S)
    with torch.cuda.device(dev):
        grad = torch.empty((P, 3), dtype=_f32, device=dev)
        rc = lib.dss_occ_backward(_lib.ptr(points), _lib.ptr(radii), _lib.ptr(vis), _lib.ptr(rs), _lib.ptr(grad_occ),
                                  _lib.ptr(first), _lib.ptr(num), N, P, S, row0, row1, gstride, float(clip),
                                  _lib.ptr(grad), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_occ_backward")
    return grad


def _backward_zbuf(idx, grad_zbuf, point_grad):
    """``DSS._C._backward_zbuf`` (ext.cpp:17): in-place scatter-add.  ``point_grad`` is either the
    reference's (P,1) z-gradient tensor or a (P,3) point-gradient tensor (z column updated)."""
    lib = _lib.load()
    idx = _lib.require_gpu(idx, "idx", _i32)
    dev = idx.device
    grad_zbuf = _lib.require_gpu(grad_zbuf, "grad_zbuf", _f32)
    if idx.dim() != 4 or idx.shape != grad_zbuf.shape:
        raise RuntimeError("idx and grad_zbuf must both be (N,H,W,K)")
    if not (point_grad.is_cuda and point_grad.dtype == _f32 and point_grad.is_contiguous() and point_grad.dim() == 2):
        raise RuntimeError("point_grad must be a contiguous float32 GPU tensor of shape (P,1) or (P,3)")
    N, H, W, K = idx.shape
    with torch.cuda.device(dev):
        if point_grad.shape[1] == 3:
            rc = lib.dss_zbuf_backward(_lib.ptr(idx), _lib.ptr(grad_zbuf), N, H, W, K, _lib.ptr(point_grad),
                                       _lib.stream_ptr(dev))
        elif point_grad.shape[1] == 1:
            tmp = torch.zeros((point_grad.shape[0], 3), dtype=_f32, device=dev)
            rc = lib.dss_zbuf_backward(_lib.ptr(idx), _lib.ptr(grad_zbuf), N, H, W, K, _lib.ptr(tmp),
                                       _lib.stream_ptr(dev))
            point_grad += tmp[:, 2:3]
        else:
            raise RuntimeError("point_grad must have 1 or 3 columns")
    _lib.check(rc, "dss_zbuf_backward")


def clip_grad_(grad_pts, clip: float):
    """In-place per-point norm clip (rasterizer.py:667-673)."""
    lib = _lib.load()
    if not (grad_pts.is_cuda and grad_pts.dtype == _f32 and grad_pts.is_contiguous()):
        raise RuntimeError("grad_pts must be a contiguous float32 GPU tensor")
    with torch.cuda.device(grad_pts.device):
        rc = lib.dss_clip_grad(_lib.ptr(grad_pts), grad_pts.shape[0], float(clip), _lib.stream_ptr(grad_pts.device))
    _lib.check(rc, "dss_clip_grad")
    return grad_pts


def splat_backward(points, radii, visible, idx, grad_occ, grad_zbuf, cloud_to_packed_first_idx,
                   num_points_per_cloud, radii_s: float, clip: float = -1.0, return_rs: bool = False):
    """Whole ``EllipticalRasterizer.backward`` (rasterizer.py:787-977) in one call -> grad (P,3)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    radii = _lib.require_gpu(radii, "radii", _f32)
    vis = _lib.require_gpu(_as_u8(visible), "visible", _u8)
    idx = _lib.require_gpu(idx, "idx", _i32)
    if grad_occ is None:  # (autograd hands over None for an unused output: only then are zeros allocated)
        grad_occ = torch.zeros(idx.shape[:3], dtype=_f32, device=dev)
    if not grad_occ.is_cuda:
        raise RuntimeError("dss_amd: grad_occ must be a GPU tensor (no CPU fallback)")
    if grad_zbuf is not None:
        grad_zbuf = _lib.require_gpu(grad_zbuf, "grad_zbuf", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, S, _, K = idx.shape
    P = points.shape[0]
    grad_occ, gstride = _pixel_strided(grad_occ, N, S, S)
    with torch.cuda.device(dev):
        grad = torch.empty((P, 3), dtype=_f32, device=dev)
        rs = torch.empty((N,), dtype=_f32, device=dev)
        ws = _lib.workspace(dev, lib.dss_splat_backward_workspace(N, P))
        rc = lib.dss_splat_backward(_lib.ptr(points), _lib.ptr(radii), _lib.ptr(vis), _lib.ptr(idx),
                                    _lib.ptr(grad_occ), _lib.ptr(grad_zbuf), _lib.ptr(first), _lib.ptr(num),
                                    N, P, S, K, gstride, float(radii_s), float(clip), _lib.ptr(grad), _lib.ptr(rs),
                                    _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_splat_backward")
    return (grad, rs) if return_rs else grad


def blend_forward(idx, qvalue, occupancy, scaler, features, return_wsum: bool = False):
    """Fused weights + NormWeightedCompositor + RGBA assembly (renderer.py:53-78).
    ``features`` is (P,C); returns (N,H,W,C+1) [and the per-pixel weight sum max(sum w, 1e-4)]."""
    lib = _lib.load()
    idx = _lib.require_gpu(idx, "idx", _i32)
    dev = idx.device
    qvalue = _lib.require_gpu(qvalue, "qvalue", _f32)
    occupancy = _lib.require_gpu(occupancy, "occupancy", _f32)
    scaler = _lib.require_gpu(scaler, "scaler", _f32)
    features = _lib.require_gpu(features, "features", _f32)
    N, H, W, K = idx.shape
    C = features.shape[1]
    with torch.cuda.device(dev):
        out = torch.empty((N, H, W, C + 1), dtype=_f32, device=dev)
        wsum = torch.empty((N, H, W), dtype=_f32, device=dev) if return_wsum else None
        rc = lib.dss_blend_forward(_lib.ptr(idx), _lib.ptr(qvalue), _lib.ptr(occupancy), _lib.ptr(scaler),
                                   _lib.ptr(features), N, H, W, K, C, _lib.ptr(out), _lib.ptr(wsum),
                                   _lib.stream_ptr(dev))
    _lib.check(rc, "dss_blend_forward")
    return (out, wsum) if return_wsum else out


def blend_backward(grad_out, idx, qvalue, scaler, num_points: int, geometry=None, wsum=None,
                   image_size: Optional[int] = None, rows: Optional[Tuple[int, int]] = None):
    """-> (grad_features (P,C), grad_occupancy (N,H,W) = strided view ``grad_out[..., C]``).

    ``geometry = (pts_screen, radii, visible, first_idx, num_points_per_cloud)`` selects the
    point-centric gather kernel (no atomics, deterministic); without it the pixel-centric scatter
    kernel is used."""
    lib = _lib.load()
    grad_out = _lib.require_gpu(grad_out, "grad_out", _f32)
    dev = grad_out.device
    idx = _lib.require_gpu(idx, "idx", _i32)
    qvalue = _lib.require_gpu(qvalue, "qvalue", _f32)
    scaler = _lib.require_gpu(scaler, "scaler", _f32)
    N, H, W, K = idx.shape
    C = grad_out.shape[-1] - 1
    with torch.cuda.device(dev):
        gf = torch.empty((num_points, C), dtype=_f32, device=dev)
        if geometry is None:
            rc = lib.dss_blend_backward_scatter(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(qvalue), _lib.ptr(scaler),
                                                N, H, W, K, C, num_points, _lib.ptr(gf), _lib.stream_ptr(dev))
            _lib.check(rc, "dss_blend_backward_scatter")
        else:
            pts, radii, vis, first, num = geometry
            pts = _lib.require_gpu(pts, "pts_screen", _f32)
            radii = _lib.require_gpu(radii, "radii", _f32)
            vis = _lib.require_gpu(_as_u8(vis), "visible", _u8)
            first = _lib.require_gpu(first, "cloud_to_packed_first_idx", _i64)
            num = _lib.require_gpu(num, "num_points_per_cloud", _i64)
            if wsum is not None:
                wsum = _lib.require_gpu(wsum, "wsum", _f32)
            S = int(image_size) if image_size is not None else W
            row0, row1 = (0, S) if rows is None else (int(rows[0]), int(rows[1]))
            if H != row1 - row0 or W != S:
                raise RuntimeError("fragment tensors must be (N, %d, %d, K)" % (row1 - row0, S))
            rc = lib.dss_blend_backward(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(qvalue), _lib.ptr(wsum),
                                        _lib.ptr(scaler), _lib.ptr(pts), _lib.ptr(radii), _lib.ptr(vis),
                                        _lib.ptr(first), _lib.ptr(num), first.shape[0], num_points, S, K, C, row0,
                                        row1, _lib.ptr(gf), _lib.stream_ptr(dev))
            _lib.check(rc, "dss_blend_backward")
    return gf, grad_out[..., C]


def backward_addr64() -> int:
    """current value of DSS_OPT_BACKWARD_ADDR64 (1 forces the 64-bit gather, which has no fused projection)"""
    return int(_lib.load().dss_get_option(_lib.OPT_BACKWARD_ADDR64))


def _band(rows, S):
    """`rows` of the fused entry points: None = whole image; (row0, row1) = contiguous band; (row0, row1, c) = of that
    band only every c-th 8-row tile row starting at row0 (tile-row-cyclic multi-GPU partition, include/dss_hip.h)."""
    if rows is None:
        return 0, int(S), 1
    return int(rows[0]), int(rows[1]), (int(rows[2]) if len(rows) > 2 else 1)


def band_rows(row0: int, row1: int, cycle: int = 1) -> int:
    """rows of a band tensor (dss_band_rows)"""
    if row1 <= row0:
        return 0
    if cycle <= 1:
        return row1 - row0
    full, rem = divmod(row1 - row0, 8 * cycle)
    return full * 8 + min(rem, 8)


def _order_state(ws, state: int, shape, order_refresh: int) -> int:
    """workspace_state of a dss_render_forward call on buffer `ws` under the cached point order (``order_refresh`` = k: the
    order saved by one call serves the next k - 1 calls on the same buffer).  The age lives on the buffer OBJECT: a new
    buffer -- or one whose last call failed and was dropped -- starts with a save."""
    if state not in (0, 1):
        return state
    if order_refresh <= 0:
        # a call without either flag sorts for itself and the library forgets the order this buffer held: the next call
        # with order_refresh > 0 must save again (e.g. two renderers of the same shape, one of them without the option)
        if getattr(ws, "_dss_order_age", None) is not None:
            ws._dss_order_age = None
        return state
    age = getattr(ws, "_dss_order_age", None)
    if age is None or age[0] != shape or age[1] + 1 >= int(order_refresh):
        ws._dss_order_age = (shape, 0)
        return state | _lib.WS_ORDER_SAVE
    ws._dss_order_age = (shape, age[1] + 1)
    return state | _lib.WS_ORDER_REUSE


def render_forward(world, normals, h, M, V, znear, zfar, cloud_to_packed_first_idx, num_points_per_cloud, features,
                   image_size: int, points_per_pixel: int, cutoff_threshold: float, depth_merging_thres: float,
                   antialiasing_sigma: float = 1.0, backface_culling: bool = False, shared_cloud: bool = False,
                   rows: Optional[Tuple[int, int]] = None, out_image: Optional[torch.Tensor] = None,
                   out_visible: Optional[torch.Tensor] = None, vr6=None, frame_normals=None, want_zbuf: bool = True,
                   workspace_state: int = 1, order_refresh: int = 0, band_outputs_only: bool = False,
                   point_outputs=None):
    """Fused forward (setup + binning + fine + blend, ``dss_render_forward``).  ``out_image`` (float32
    (N,rows,S,C+1), 16-byte aligned) / ``out_visible`` (uint8 (P,)) let the caller place these two outputs
    in its own buffer (the multi-GPU step points them into one all-gather send buffer).  ``features`` are the
    packed (P,C) features.  Returns a dict with everything the separate calls produce:
    ``pts_screen, ellipse_params, radii, scaler, cutoff_threshold, valid, idx, zbuf, qvalue, occupancy,
    visible, image, wsum``.  ``want_zbuf=False`` skips the depth plane (``zbuf`` is None): the fused backward never reads it.
    ``workspace_state``: 1 = DSS_WS_CLEAN (default: cached zero-initialised workspace, no memset launch), 0 = DSS_WS_UNKNOWN
    (memset + both launches, lists stay in place), 2 = DSS_WS_BINNED (after a state-0 call with the same inputs: repeat only
    the fine + blend launch; profiling / timing of the dominant kernel).
    ``order_refresh`` = k > 0: renderer-owned cached point order (DSS_WS_ORDER_SAVE / DSS_WS_ORDER_REUSE, above 2M points):
    the screen-cell order that the binning sorts the points into is kept in the workspace of this problem size and reused by
    the next k - 1 calls on it, which then skip the sort (same outputs bit for bit; a stale order only costs locality).
    ``band_outputs_only`` (DSS_WS_BAND_OUTPUTS; with ``rows`` only): ``ellipse_params``, ``scaler`` and ``cutoff_threshold`` are
    written for the splats that meet the band only (zero elsewhere); everything else is unchanged -- what a multi-GPU
    rank asks for: it needs every point's position and radii for the backward, but bins an eighth of the cloud.
    ``point_outputs`` = (ellipse (P,3), scaler (P,), cutoff (P,)): caller-owned buffers for those three outputs."""
    lib = _lib.load()
    world = _lib.require_gpu(world, "world", _f32)
    dev = world.device
    normals = _lib.require_gpu(normals, "normals", _f32)
    h = _lib.require_gpu(h, "h", _f32)
    M = _lib.require_gpu(M, "M", _f32)
    V = _lib.require_gpu(V, "V", _f32)
    znear = _lib.require_gpu(znear, "znear", _f32)
    zfar = _lib.require_gpu(zfar, "zfar", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    features = _lib.require_gpu(features, "features", _f32)
    N, Pw = first.shape[0], world.shape[0]
    if tuple(M.shape) != (N, 4, 4) or tuple(V.shape) != (N, 4, 4) or znear.numel() != N or zfar.numel() != N:
        raise RuntimeError("camera tensors must be M,V (N,4,4) and znear,zfar (N,) with N=%d" % N)
    if normals.shape != world.shape:
        raise RuntimeError("normals must match world points")
    P = N * Pw if shared_cloud else Pw
    if features.shape[0] != P:
        raise RuntimeError("features must be packed (P,C) with P=%d, got %s" % (P, tuple(features.shape)))
    per_point = h.numel() == Pw and not (h.numel() == N and Pw == N)
    packed_h = (not per_point) and shared_cloud and N > 1 and h.numel() == P    # one value per (camera, point) pair
    if not per_point and not packed_h and h.numel() != N:
        raise RuntimeError("h must have %d (per point), %d (per cloud) or, for a shared cloud, %d (per packed point) entries" % (Pw, N, P))
    S, K, C = int(image_size), int(points_per_pixel), features.shape[1]
    row0, row1, cyc = _band(rows, S)
    nr = band_rows(row0, row1, cyc)
    e = lambda *shape, dtype=_f32: torch.empty(shape, dtype=dtype, device=dev)
    if nr == 0:
        # empty row band (multi-GPU rank without rows): only the per-point setup runs; no fragments, nothing visible
        o = point_setup(world, normals, h, M, V, znear, zfar, first, num, S, cutoff_threshold, antialiasing_sigma,
                        backface_culling, shared_cloud, vr6=vr6, frame_normals=frame_normals)
        vis = torch.zeros(P, dtype=_u8, device=dev) if out_visible is None else out_visible.zero_()
        o.update(idx=e(N, 0, S, K, dtype=_i32), zbuf=e(N, 0, S, K), qvalue=e(N, 0, S, K), occupancy=e(N, 0, S),
                 image=e(N, 0, S, C + 1) if out_image is None else out_image, wsum=e(N, 0, S), visible=vis.view(torch.bool))
        return o
    with torch.cuda.device(dev):
        # band_outputs_only: the library writes ellipse / scaler / cutoff only for the splats that meet the band; the rest of
        # those three arrays is handed out ZERO-filled, never as uninitialised device memory (one fill of 20 P bytes)
        if point_outputs is not None:
            # caller-owned (ellipse (P,3), scaler (P,), cutoff (P,)): a multi-GPU step zero-fills them ONCE and hands them to
            # every forward (entries of splats outside the band then hold zeros or the value of an earlier step: defined)
            ell, sca, cut = point_outputs
            if tuple(ell.shape) != (P, 3) or tuple(sca.shape) != (P,) or tuple(cut.shape) != (P,) or \
                    any(t.dtype != _f32 or not t.is_contiguous() or t.device != dev for t in (ell, sca, cut)):
                raise RuntimeError("point_outputs must be contiguous float32 (P,3), (P,), (P,) tensors on the device")
        elif band_outputs_only and (int(workspace_state) & 0xf) != 2:
            z3 = torch.zeros(P * 5, dtype=_f32, device=dev)
            ell, sca, cut = z3[:P * 3].view(P, 3), z3[P * 3:P * 4], z3[P * 4:]
        else:
            ell, sca, cut = e(P, 3), e(P), e(P)
        o = dict(pts_screen=e(P, 3), ellipse_params=ell, radii=e(P, 2), scaler=sca, cutoff_threshold=cut,
                 idx=e(N, nr, S, K, dtype=_i32), zbuf=e(N, nr, S, K) if want_zbuf else None, qvalue=e(N, nr, S, K),
                 occupancy=e(N, nr, S), image=e(N, nr, S, C + 1) if out_image is None else out_image, wsum=e(N, nr, S))
        valid = e(P, dtype=_u8)
        vis = e(P, dtype=_u8) if out_visible is None else out_visible
        img = o["image"]
        if tuple(img.shape) != (N, nr, S, C + 1) or img.dtype != _f32 or img.data_ptr() % 16 \
                or img.stride(3) != 1 or img.stride(2) != C + 1 or img.stride(0) % 4 or img.stride(1) % 4 \
                or tuple(vis.shape) != (P,) or vis.dtype != _u8:
            raise RuntimeError("out_image must be float32 (N,rows,S,C+1), 16-byte aligned, with contiguous rows "
                               "(any camera / row strides that are multiples of 4 floats); out_visible uint8 (P,)")
        _keep, vr_p, fn_p = _aniso_args(vr6, frame_normals, Pw)
        # dedicated zero-initialised buffer per problem size: the library keeps it clean (no memset launch)
        tag = ("render_forward" if (int(workspace_state) & 0xf) == 1 else "render_forward_binned", N, P, S)
        ws = _lib.clean_workspace(dev, tag, lib.dss_render_forward_workspace(N, P, S, K))
        state = _order_state(ws, int(workspace_state), (N, P, S), order_refresh)
        if band_outputs_only and (int(workspace_state) & 0xf) != 2:
            state |= _lib.WS_BAND_OUTPUTS
        rc = lib.dss_render_forward(
            _lib.ptr(world), _lib.ptr(normals), _lib.ptr(h) if (per_point or packed_h) else None, None if per_point else _lib.ptr(h),
            vr_p, fn_p, _lib.ptr(M), _lib.ptr(V), _lib.ptr(znear), _lib.ptr(zfar), _lib.ptr(first), _lib.ptr(num), N, P,
            int(shared_cloud), int(backface_culling), S, K, float(cutoff_threshold), float(antialiasing_sigma),
            float(depth_merging_thres), row0, row1, cyc, _lib.ptr(features), C, _lib.ptr(o["pts_screen"]),
            _lib.ptr(o["ellipse_params"]), _lib.ptr(o["radii"]), _lib.ptr(o["scaler"]), _lib.ptr(o["cutoff_threshold"]),
            _lib.ptr(valid), _lib.ptr(o["idx"]), _lib.ptr(o["zbuf"]), _lib.ptr(o["qvalue"]), _lib.ptr(o["occupancy"]),
            _lib.ptr(vis), _lib.ptr(img), int(img.stride(0)), int(img.stride(1)), _lib.ptr(o["wsum"]), _lib.ptr(ws),
            ws.numel(), state, _lib.stream_ptr(dev))
        if rc:
            _lib.drop_clean_workspace(dev, tag)
    _lib.check(rc, "dss_render_forward")
    o["valid"], o["visible"] = valid.view(torch.bool), vis.view(torch.bool)
    return o


def render_backward(grad_out, idx, qvalue, wsum, scaler, points, radii, visible, cloud_to_packed_first_idx,
                    num_points_per_cloud, radii_s: float, clip: float = -1.0, with_features: bool = True,
                    return_rs: bool = False, image_size: Optional[int] = None,
                    rows: Optional[Tuple[int, int]] = None, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                    gather_only_rs: Optional[torch.Tensor] = None, project=None,
                    grad_out_full: Optional[torch.Tensor] = None, grad_occ_full: Optional[torch.Tensor] = None):
    """Fused backward of renderer + rasterizer (blend backward + median radius + occupancy backward +
    clip) -> (grad_features (P,C) or None, grad_pts_screen (P,3)).  With ``rows`` (multi-GPU band) pass the
    union of the visibility flags and ``clip <= 0``; the results are the band's partial sums.
    ``project=(world (P,3), M (N,4,4))`` also fuses `project_backward` into the launch: the second result is then the
    WORLD-space position gradient (clouds that are not shared between cameras, whole image, 3 feature channels).
    ``gather_only_rs`` = the ``rs`` a preceding identical call returned (and ``out`` = its outputs): re-runs only the
    second stage, the gather kernel (``dss_render_backward_gather``; per-kernel timing).
    ``grad_out_full`` (N,S,S,C+1) with ``rows``: OWNER mode of the band (``dss_render_backward_owned``) -- the position
    gradient of every (camera, point) pair is computed completely, over its whole search window in the full image gradient,
    by the rank whose band holds the image row of the point's centre (zeros on the other ranks), so that clip and projection
    can precede the reduction over the ranks; the feature gradients stay partial sums of the band.
    ``grad_occ_full`` (N,S,S) with ``rows``: the same owner mode fed by the dense plane of the occupancy gradient alone
    (``dss_render_backward_owned_plane``) -- what the ranks of a step with a band-local loss all-gather (`gather_rows`)."""
    lib = _lib.load()
    grad_out = _lib.require_gpu(grad_out, "grad_out", _f32)
    dev = grad_out.device
    idx = _lib.require_gpu(idx, "idx", _i32)
    qvalue = _lib.require_gpu(qvalue, "qvalue", _f32)
    scaler = _lib.require_gpu(scaler, "scaler", _f32)
    points = _lib.require_gpu(points, "points", _f32)
    radii = _lib.require_gpu(radii, "radii", _f32)
    vis = _lib.require_gpu(_as_u8(visible), "visible", _u8)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    if wsum is not None:
        wsum = _lib.require_gpu(wsum, "wsum", _f32)
    N, H, W, K = idx.shape
    C = grad_out.shape[-1] - 1
    P = points.shape[0]
    S = int(image_size) if image_size is not None else W
    row0, row1, cyc = _band(rows, S)
    if W != S or H != band_rows(row0, row1, cyc) or tuple(grad_out.shape[:3]) != (N, H, W):
        raise RuntimeError("render_backward needs idx (N,rows,S,K) and grad_out (N,rows,S,C+1)")
    with torch.cuda.device(dev):
        if out is not None:  # caller-provided (P,C) / (P,3) float32 views, e.g. slices of one all-reduce bucket
            gf, gp = out
            if tuple(gf.shape) != (P, C) or tuple(gp.shape) != (P, 3) or gf.dtype != _f32 or gp.dtype != _f32 \
                    or not gf.is_contiguous() or not gp.is_contiguous():
                raise RuntimeError("out must be contiguous float32 tensors of shape (P,C) and (P,3)")
        else:
            gf = torch.empty((P, C), dtype=_f32, device=dev) if with_features else None
            gp = torch.empty((P, 3), dtype=_f32, device=dev)
        if H == 0:  # empty row band: zero partial sums (the search radius still follows the global visibility)
            if gf is not None:
                gf.zero_()
            gp.zero_()
            if return_rs:
                return gf, gp, backward_radius(radii, vis, first, num, radii_s)
            return gf, gp
        rs = torch.empty((N,), dtype=_f32, device=dev) if gather_only_rs is None else \
            _lib.require_gpu(gather_only_rs, "gather_only_rs", _f32)
        if gather_only_rs is not None and out is None:
            raise RuntimeError("gather_only_rs needs out= (the gradients the full call zero-filled)")
        w_p = m_p = None
        if project is not None:
            w_t, m_t = _lib.require_gpu(project[0], "world", _f32), _lib.require_gpu(project[1], "M", _f32)
            if tuple(w_t.shape) != (P, 3) or tuple(m_t.shape) != (N, 4, 4):
                raise RuntimeError("project=(world (P,3), M (N,4,4)) with P=%d N=%d, got %s %s"
                                   % (P, N, tuple(w_t.shape), tuple(m_t.shape)))
            w_p, m_p = _lib.ptr(w_t), _lib.ptr(m_t)
        ws = _lib.workspace(dev, lib.dss_render_backward_workspace(N, P, S))
        if grad_occ_full is not None:
            if project is not None or gather_only_rs is not None or grad_out_full is not None:
                raise RuntimeError("grad_occ_full (owner mode of a band) excludes project=, gather_only_rs= and grad_out_full=")
            plane = _lib.require_gpu(grad_occ_full, "grad_occ_full", _f32)
            if tuple(plane.shape) != (N, S, S):
                raise RuntimeError("grad_occ_full must be (N,S,S) = %s, got %s" % ((N, S, S), tuple(plane.shape)))
            rc = lib.dss_render_backward_owned_plane(
                _lib.ptr(grad_out), _lib.ptr(plane), _lib.ptr(idx), _lib.ptr(qvalue), _lib.ptr(wsum), _lib.ptr(scaler),
                _lib.ptr(points), _lib.ptr(radii), _lib.ptr(vis), _lib.ptr(first), _lib.ptr(num), N, P, S, K, C, row0, row1, cyc,
                float(radii_s), float(clip), _lib.ptr(gf), _lib.ptr(gp), _lib.ptr(rs), _lib.ptr(ws), ws.numel(),
                _lib.stream_ptr(dev))
            _lib.check(rc, "dss_render_backward_owned_plane")
            return (gf, gp, rs) if return_rs else (gf, gp)
        if grad_out_full is not None:
            if project is not None or gather_only_rs is not None:
                raise RuntimeError("grad_out_full (owner mode of a band) excludes project= and gather_only_rs=")
            full = _lib.require_gpu(grad_out_full, "grad_out_full", _f32)
            if tuple(full.shape) != (N, S, S, C + 1) or not full.is_contiguous():
                raise RuntimeError("grad_out_full must be contiguous (N,S,S,C+1)")
            rc = lib.dss_render_backward_owned(
                _lib.ptr(grad_out), _lib.ptr(full), _lib.ptr(idx), _lib.ptr(qvalue), _lib.ptr(wsum), _lib.ptr(scaler),
                _lib.ptr(points), _lib.ptr(radii), _lib.ptr(vis), _lib.ptr(first), _lib.ptr(num), N, P, S, K, C, row0, row1, cyc,
                float(radii_s), float(clip), _lib.ptr(gf), _lib.ptr(gp), _lib.ptr(rs), _lib.ptr(ws), ws.numel(),
                _lib.stream_ptr(dev))
            _lib.check(rc, "dss_render_backward_owned")
            return (gf, gp, rs) if return_rs else (gf, gp)
        entry = lib.dss_render_backward if gather_only_rs is None else lib.dss_render_backward_gather
        rc = entry(_lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(qvalue), _lib.ptr(wsum),
                   _lib.ptr(scaler), _lib.ptr(points), _lib.ptr(radii), _lib.ptr(vis),
                   _lib.ptr(first), _lib.ptr(num), N, P, S, K, C, row0, row1, cyc, float(radii_s),
                   float(clip), _lib.ptr(gf), _lib.ptr(gp), _lib.ptr(rs), w_p, m_p, _lib.ptr(ws), ws.numel(),
                   _lib.stream_ptr(dev))
    _lib.check(rc, "dss_render_backward")
    return (gf, gp, rs) if return_rs else (gf, gp)


def gather_rows(src, row_pos, n_images: int, rows: int, row_floats: int, out=None):
    """``dss_gather_rows``: rows of an all-gathered (position, camera, row_floats) float32 buffer put in image order ->
    dense (n_images, rows, row_floats); ``row_pos`` int32 (rows,) = position of image row r in ``src``."""
    lib = _lib.load()
    src = _lib.require_gpu(src, "src", _f32)
    dev = src.device
    row_pos = _lib.require_gpu(row_pos, "row_pos", _i32)
    if row_pos.numel() != rows or src.numel() % (n_images * row_floats):
        raise RuntimeError("gather_rows: row_pos must have %d entries and src whole (camera, row) slices" % rows)
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty((n_images, rows, row_floats), dtype=_f32, device=dev)
        elif out.numel() != n_images * rows * row_floats or out.dtype != _f32 or not out.is_contiguous():
            raise RuntimeError("gather_rows: out must be a contiguous float32 tensor of %d elements" % (n_images * rows * row_floats))
        rc = lib.dss_gather_rows(_lib.ptr(src), _lib.ptr(row_pos), int(n_images), int(rows), int(row_floats), _lib.ptr(out),
                                 _lib.stream_ptr(dev))
    _lib.check(rc, "dss_gather_rows")
    return out


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_SWITCH = _NoSwitch()


def _on_device(dev):
    """`torch.cuda.device(dev)` only when `dev` is not already the current device (the context manager costs ~5 us)"""
    return _NO_SWITCH if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


class FusedPlan:
    """Host-lean form of ``render_forward`` + ``render_backward`` for one problem shape (the drop-in renderer's hot path,
    DSS/core/renderer.py:36-82 + rasterizer.py:584-664): ONE arena allocation per forward carved into the 13 output
    tensors (only the ones somebody reads become tensor objects), the workspace looked up once, the ~45 arguments of the
    two C entry points prebuilt with only the pointers refreshed per call.  Same kernels, same results; the general
    functions above stay the reference for argument checking (a plan is only built from inputs they would accept)."""

    _ALIGN = 256

    def __init__(self, device, N, Pw, P, S, K, C, shared, per_point_h, aniso, backface, cutoff, sigma, thr, want_zbuf=True):
        self.lib = _lib.load()
        self.dev, self.N, self.Pw, self.P, self.S, self.K, self.C = device, N, Pw, P, S, K, C
        # per_point_h: 0 = one h per cloud, 1 = per world point, 2 = per PACKED point (shared cloud, cameras that cull differently)
        self.shared, self.per_point_h, self.aniso = bool(shared), int(per_point_h), bool(aniso)
        off = [0]

        def take(nbytes):
            o = off[0]
            off[0] += (int(nbytes) + self._ALIGN - 1) // self._ALIGN * self._ALIGN
            return o
        px = N * S * S
        self.layout = {}   # name -> (offset, bytes, dtype, shape)
        for name, nbytes, dt, shape in (
                ("image", px * (C + 1) * 4, _f32, (N, S, S, C + 1)), ("pts_screen", P * 12, _f32, (P, 3)),
                ("ellipse_params", P * 12, _f32, (P, 3)), ("radii", P * 8, _f32, (P, 2)), ("scaler", P * 4, _f32, (P,)),
                ("cutoff_threshold", P * 4, _f32, (P,)), ("valid", P, _u8, (P,)), ("visible", P, _u8, (P,)),
                ("idx", px * K * 4, _i32, (N, S, S, K)), ("qvalue", px * K * 4, _f32, (N, S, S, K)),
                ("occupancy", px * 4, _f32, (N, S, S)), ("wsum", px * 4, _f32, (N, S, S))) + (
                (("zbuf", px * K * 4, _f32, (N, S, S, K)),) if want_zbuf else ()):
            self.layout[name] = (take(nbytes), int(nbytes), dt, shape)
        self.total = off[0]
        self.want_zbuf = want_zbuf
        self.order_refresh = 0   # k > 0: cached point order, see render_forward (set by the rasterizer that owns the plan)
        self.force_state = None  # workspace_state passed as is (a caller that captures the saving and the reusing call as graphs)
        self.fwd_ws_bytes = self.lib.dss_render_forward_workspace(N, P, S, K)
        self.bwd_ws_bytes = self.lib.dss_render_backward_workspace(N, P, S)
        self.tag = ("render_forward", N, P, S)
        self.consts = (int(shared), int(backface), S, K, float(cutoff), float(sigma), float(thr))
        self._o = {k: v[0] for k, v in self.layout.items()}

    @staticmethod
    def lean_input(t, dtype=_f32):
        return t is not None and t.is_cuda and t.dtype == dtype and t.is_contiguous()

    def view(self, arena, name):
        o, n, dt, shape = self.layout[name]
        return arena[o:o + n].view(dt).view(shape)

    def image(self, arena):
        """the (N,S,S,C+1) image at offset 0 of the arena"""
        n, shape = self.layout["image"][1], self.layout["image"][3]
        return arena[:n].view(_f32).view(shape)

    def forward(self, world, normals, h, M, V, znear, zfar, first, num, feats, vr6=None, frame_n=None, ws=None):
        """-> arena (uint8): every output of dss_render_forward at its offset (see `view`).  `ws`: a caller-owned forward
        workspace under the DSS_WS_CLEAN contract (zero-filled once, `fwd_ws_bytes` long) instead of the cached per-stream
        one -- a captured graph bakes the workspace address into its launches and must own the buffer."""
        lib, dev, o = self.lib, self.dev, self._o
        N, P, S, K, C = self.N, self.P, self.S, self.K, self.C
        shared, backface, _, _, cutoff, sigma, thr = self.consts
        own_ws = ws is not None
        with _on_device(dev):
            arena = torch.empty(self.total, dtype=_u8, device=dev)
            if not own_ws:
                ws = _lib.clean_workspace(dev, self.tag, self.fwd_ws_bytes)
            b = arena.data_ptr()
            hp = h.data_ptr()
            rc = lib.dss_render_forward(
                world.data_ptr(), normals.data_ptr(), hp if self.per_point_h else None, None if self.per_point_h == 1 else hp,
                None if vr6 is None else vr6.data_ptr(), None if frame_n is None else frame_n.data_ptr(),
                M.data_ptr(), V.data_ptr(), znear.data_ptr(), zfar.data_ptr(), first.data_ptr(), num.data_ptr(), N, P,
                shared, backface, S, K, cutoff, sigma, thr, 0, S, 1, feats.data_ptr(), C,
                b + o["pts_screen"], b + o["ellipse_params"], b + o["radii"], b + o["scaler"], b + o["cutoff_threshold"],
                b + o["valid"], b + o["idx"], (b + o["zbuf"]) if self.want_zbuf else None, b + o["qvalue"], b + o["occupancy"],
                b + o["visible"], b + o["image"], 0, 0, b + o["wsum"], ws.data_ptr(), ws.numel(),
                self.force_state if self.force_state is not None else _order_state(ws, 1, (N, P, S), self.order_refresh),
                torch.cuda.current_stream(dev).cuda_stream)
            if rc:
                if own_ws:
                    ws.zero_()   # (a failed call may leave counters behind: restore the contract's state)
                else:
                    _lib.drop_clean_workspace(dev, self.tag)
        _lib.check(rc, "dss_render_forward")
        return arena

    def backward(self, arena, g_image, first, num, radii_s, clip, world=None, M=None, ws=None):
        """-> (grad_features (P,C), grad_pts (P,3)) -- world-space position gradients when (world, M) are given (fused
        projection, see render_backward).  `ws`: a caller-owned backward workspace (`bwd_ws_bytes`), see `forward`."""
        lib, dev, o = self.lib, self.dev, self._o
        N, P, S, K, C = self.N, self.P, self.S, self.K, self.C
        with _on_device(dev):
            if C == 3:
                both = torch.empty((2, P, 3), dtype=_f32, device=dev)
                gf, gp = both[0], both[1]
            else:
                gf = torch.empty((P, C), dtype=_f32, device=dev)
                gp = torch.empty((P, 3), dtype=_f32, device=dev)
            rs = torch.empty((N,), dtype=_f32, device=dev)
            if ws is None:
                ws = _lib.workspace(dev, self.bwd_ws_bytes)
            b = arena.data_ptr()
            rc = lib.dss_render_backward(
                g_image.data_ptr(), b + o["idx"], b + o["qvalue"], b + o["wsum"], b + o["scaler"], b + o["pts_screen"],
                b + o["radii"], b + o["visible"], first.data_ptr(), num.data_ptr(), N, P, S, K, C, 0, S, 1, float(radii_s),
                float(clip), gf.data_ptr(), gp.data_ptr(), rs.data_ptr(), None if world is None else world.data_ptr(),
                None if M is None else M.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "dss_render_backward")
        return gf, gp


def _aniso_args(vr6, frame_normals, Pw):
    """(ptr(vr6), ptr(frame_normals)) of the anisotropic source variance, or (None, None)."""
    if vr6 is None:
        return None, None, None
    vr6 = _lib.require_gpu(vr6, "vr6", _f32)
    fn = _lib.require_gpu(frame_normals, "frame_normals", _f32)
    if tuple(vr6.shape) != (Pw, 6) or tuple(fn.shape) != (Pw, 3):
        raise RuntimeError("anisotropic mode needs vr6 (%d,6) and frame_normals (%d,3)" % (Pw, Pw))
    return (vr6, fn), _lib.ptr(vr6), _lib.ptr(fn)


def local_frames(points, knn_idx, cloud_to_packed_first_idx, num_points_per_cloud, return_curvature: bool = False):
    """PCA frames of the K-neighbourhoods (``knn_idx`` from ``knn_points``, self included) -> anisotropic source
    variance ``vr6 (P,6)`` (xx,xy,xz,yy,yz,zz) and frame normals ``(P,3)`` (rasterizer.py:256-291 +
    mathHelper.py:34-92 with neighborhood_size = 8)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    knn_idx = _lib.require_gpu(knn_idx, "knn_idx", _i64)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    P, K = knn_idx.shape
    if points.shape[0] != P:
        raise RuntimeError("knn_idx must be (P,K) for the P packed points")
    with torch.cuda.device(dev):
        vr6 = torch.empty((P, 6), dtype=_f32, device=dev)
        fn = torch.empty((P, 3), dtype=_f32, device=dev)
        cv = torch.empty((P, 3), dtype=_f32, device=dev) if return_curvature else None
        rc = lib.dss_local_frames(_lib.ptr(points), _lib.ptr(knn_idx), _lib.ptr(first), _lib.ptr(num), first.shape[0], P,
                                  int(K), _lib.ptr(vr6), _lib.ptr(fn), _lib.ptr(cv), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_local_frames")
    return (vr6, fn, cv) if return_curvature else (vr6, fn)


def point_setup(world, normals, h, M, V, znear, zfar, cloud_to_packed_first_idx, num_points_per_cloud,
                image_size: int, cutoff_threshold: float, antialiasing_sigma: float = 1.0,
                backface_culling: bool = False, shared_cloud: bool = False, vr6=None, frame_normals=None):
    """Fused culling + projection + EWA per-point setup (rasterizer.py:183-254, 443-565, 614).

    ``h`` is either per point ``(Pw,)`` or per cloud ``(N,)``; with ``vr6`` / ``frame_normals`` (``local_frames``)
    the anisotropic source variance is used instead and ``h`` is ignored.  Returns a dict with
    ``pts_screen (P,3), ellipse_params (P,3), radii (P,2), scaler (P,), cutoff_threshold (P,),
    valid bool (P,)``.
    """
    lib = _lib.load()
    world = _lib.require_gpu(world, "world", _f32)
    dev = world.device
    normals = _lib.require_gpu(normals, "normals", _f32)
    h = _lib.require_gpu(h, "h", _f32)
    M = _lib.require_gpu(M, "M", _f32)
    V = _lib.require_gpu(V, "V", _f32)
    znear = _lib.require_gpu(znear, "znear", _f32)
    zfar = _lib.require_gpu(zfar, "zfar", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, Pw = first.shape[0], world.shape[0]
    if tuple(M.shape) != (N, 4, 4) or tuple(V.shape) != (N, 4, 4) or znear.numel() != N or zfar.numel() != N:
        raise RuntimeError("camera tensors must be M,V (N,4,4) and znear,zfar (N,) with N=%d" % N)
    if normals.shape != world.shape:
        raise RuntimeError("normals must match world points")
    P = N * Pw if shared_cloud else Pw
    per_point = h.numel() == Pw and not (h.numel() == N and Pw == N)
    packed_h = (not per_point) and shared_cloud and N > 1 and h.numel() == P    # one value per (camera, point) pair
    if not per_point and not packed_h and h.numel() != N:
        raise RuntimeError("h must have %d (per point), %d (per cloud) or, for a shared cloud, %d (per packed point) entries" % (Pw, N, P))
    with torch.cuda.device(dev):
        out = dict(pts_screen=torch.empty((P, 3), dtype=_f32, device=dev),
                   ellipse_params=torch.empty((P, 3), dtype=_f32, device=dev),
                   radii=torch.empty((P, 2), dtype=_f32, device=dev),
                   scaler=torch.empty((P,), dtype=_f32, device=dev),
                   cutoff_threshold=torch.empty((P,), dtype=_f32, device=dev))
        valid = torch.empty((P,), dtype=_u8, device=dev)
        _keep, vr_p, fn_p = _aniso_args(vr6, frame_normals, Pw)
        rc = lib.dss_point_setup(_lib.ptr(world), _lib.ptr(normals), _lib.ptr(h) if (per_point or packed_h) else None,
                                 None if per_point else _lib.ptr(h), vr_p, fn_p, _lib.ptr(M), _lib.ptr(V), _lib.ptr(znear),
                                 _lib.ptr(zfar), _lib.ptr(first), _lib.ptr(num), N, P, int(shared_cloud),
                                 int(backface_culling), int(image_size), float(cutoff_threshold),
                                 float(antialiasing_sigma), _lib.ptr(out["pts_screen"]),
                                 _lib.ptr(out["ellipse_params"]), _lib.ptr(out["radii"]), _lib.ptr(out["scaler"]),
                                 _lib.ptr(out["cutoff_threshold"]), _lib.ptr(valid), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_point_setup")
    out["valid"] = valid.view(torch.bool)
    return out


def project_backward(world, M, V, cloud_to_packed_first_idx, num_points_per_cloud, grad_screen, valid,
                     shared_cloud: bool = False, clip: float = -1.0, grad_features=None, out=None):
    """grad of (NDC x, NDC y, view z) w.r.t. the world points -> (Pw,3).  ``clip > 0`` applies the per-point norm
    clip of ``clip_grad_`` to ``grad_screen`` on the fly (multi-GPU: the clip comes after the all-reduce).
    ``grad_features`` (P,C): also sums the per-camera feature gradients of a shared cloud over its cameras in the same
    launch (``dss_project_backward_features``) -> (grad_world (Pw,3), grad_features_world (Pw,C)).
    ``out`` = (grad_world, grad_features_world) to write into (contiguous float32, e.g. two views of one all-reduce
    buffer)."""
    lib = _lib.load()
    world = _lib.require_gpu(world, "world", _f32)
    dev = world.device
    M = _lib.require_gpu(M, "M", _f32)
    V = _lib.require_gpu(V, "V", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    grad_screen = _lib.require_gpu(grad_screen, "grad_screen", _f32)
    vis = _lib.require_gpu(_as_u8(valid), "valid", _u8)
    N, Pw = first.shape[0], world.shape[0]
    with torch.cuda.device(dev):
        gw = torch.empty((Pw, 3), dtype=_f32, device=dev) if out is None else out[0]
        if out is not None and (tuple(gw.shape) != (Pw, 3) or gw.dtype != _f32 or not gw.is_contiguous()):
            raise RuntimeError("out[0] must be a contiguous float32 (Pw,3) tensor")
        if grad_features is not None:
            gfeat = _lib.require_gpu(grad_features, "grad_features", _f32)
            P = N * Pw if shared_cloud else Pw
            if gfeat.dim() != 2 or gfeat.shape[0] != P:
                raise RuntimeError("grad_features must be packed (P,C) with P=%d, got %s" % (P, tuple(gfeat.shape)))
            C = gfeat.shape[1]
            gfw = torch.empty((Pw, C), dtype=_f32, device=dev) if out is None else out[1]
            if out is not None and (tuple(gfw.shape) != (Pw, C) or gfw.dtype != _f32 or not gfw.is_contiguous()):
                raise RuntimeError("out[1] must be a contiguous float32 (Pw,C) tensor")
            rc = lib.dss_project_backward_features(_lib.ptr(world), _lib.ptr(M), _lib.ptr(V), _lib.ptr(first), _lib.ptr(num), N,
                                                   Pw, int(shared_cloud), _lib.ptr(grad_screen), _lib.ptr(vis), float(clip),
                                                   _lib.ptr(gw), _lib.ptr(gfeat), C, _lib.ptr(gfw), _lib.stream_ptr(dev))
            _lib.check(rc, "dss_project_backward_features")
            return gw, gfw
        rc = lib.dss_project_backward(_lib.ptr(world), _lib.ptr(M), _lib.ptr(V), _lib.ptr(first), _lib.ptr(num), N,
                                      Pw, int(shared_cloud), _lib.ptr(grad_screen), _lib.ptr(vis), float(clip),
                                      _lib.ptr(gw), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_project_backward")
    return gw


def knn_kth_sqdist(points, cloud_to_packed_first_idx, num_points_per_cloud, K: int = 7, radius=None):
    """K-th smallest squared distance of every point to its own cloud (self included) -> (P,).
    Exact grid search in HIP; replaces frnn.frnn_grid_points / pytorch3d.ops.knn_points for the
    variance-scale statistic (rasterizer.py:310-321, 366-383).  ``radius`` > 0 (``dss_knn_kth_sqdist_radius``): the
    fixed-radius semantics of the reference's DEFAULT search, ``frnn_grid_points(K, r = frnn_radius = 0.2)`` -- the largest of
    the K - 1 neighbour distances that lie within ``radius``, -1 for a point that has no neighbour there."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, P = first.shape[0], points.shape[0]
    with torch.cuda.device(dev):
        out = torch.empty((P,), dtype=_f32, device=dev)
        ws = _lib.workspace(dev, lib.dss_knn_workspace(N, P))
        if radius is not None and radius > 0:
            rc = lib.dss_knn_kth_sqdist_radius(_lib.ptr(points), _lib.ptr(first), _lib.ptr(num), N, P, int(K), float(radius),
                                               _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        else:
            rc = lib.dss_knn_kth_sqdist(_lib.ptr(points), _lib.ptr(first), _lib.ptr(num), N, P, int(K), _lib.ptr(out),
                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_knn_kth_sqdist")
    return out


def knn_points(points, cloud_to_packed_first_idx, num_points_per_cloud, K: int):
    """Self kNN of packed clouds -> (dists (P,K) squared, idx (P,K) int64 cloud-local), ascending, the point itself
    first, zero-padded for clouds with fewer than K points: the packed form of
    ``pytorch3d.ops.knn_points(p, p, lengths, lengths, K)`` used by the regularisers (losses.py:157-180)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, P = first.shape[0], points.shape[0]
    with torch.cuda.device(dev):
        dists = torch.empty((P, int(K)), dtype=_f32, device=dev)
        idx = torch.empty((P, int(K)), dtype=_i64, device=dev)
        ws = _lib.workspace(dev, lib.dss_knn_workspace(N, P))
        rc = lib.dss_knn_points(_lib.ptr(points), _lib.ptr(first), _lib.ptr(num), N, P, int(K), _lib.ptr(dists),
                                _lib.ptr(idx), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_knn_points")
    return dists, idx


def cloud_mean_clamp(values, cloud_to_packed_first_idx, num_points_per_cloud, scale: float, lo: float, hi: float,
                     fallback: float, min_points: int):
    """Per-cloud clamp(mean(values*scale), lo, hi) -> (N,), deterministic."""
    lib = _lib.load()
    values = _lib.require_gpu(values, "values", _f32)
    dev = values.device
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N = first.shape[0]
    with torch.cuda.device(dev):
        out = torch.empty((N,), dtype=_f32, device=dev)
        rc = lib.dss_cloud_mean_clamp(_lib.ptr(values), _lib.ptr(first), _lib.ptr(num), N, float(scale), float(lo),
                                      float(hi), float(fallback), int(min_points), _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_cloud_mean_clamp")
    return out


def renderable_mean_clamp(values, world, V, znear, zfar, cloud_to_packed_first_idx, num_points_per_cloud, shared_cloud: bool,
                          scale: float, lo: float, hi: float, fallback: float, min_points: int):
    """``cloud_mean_clamp`` under the reference's depth culling for MASKED clouds (``dss_renderable_mean_clamp``): per camera
    the mean over the points it keeps (view z in [znear, zfar]), divided by the LARGEST kept count of the batch like the
    reference's mean over the padded clouds (rasterizer.py:183-217, 320-326) -> (N,).  ``values``: (Pw,) one per world point,
    or (N, Pw) per (camera, point) from `knn_kth_sqdist_view` (a shared cloud in the reference's exact order)."""
    lib = _lib.load()
    values = _lib.require_gpu(values, "values", _f32)
    dev = values.device
    world = _lib.require_gpu(world, "world", _f32)
    V = _lib.require_gpu(V, "V", _f32)
    znear = _lib.require_gpu(znear, "znear", _f32)
    zfar = _lib.require_gpu(zfar, "zfar", _f32)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    N, Pw = V.shape[0], world.shape[0]
    per_cam = values.dim() == 2
    if (tuple(values.shape) != ((N, Pw) if per_cam else (Pw,))) or first.shape[0] != N or znear.numel() != N or zfar.numel() != N:
        raise RuntimeError("renderable_mean_clamp: values (Pw,) or (N,Pw); V, znear, zfar, first_idx, num_points per camera")
    with torch.cuda.device(dev):
        out = torch.empty((N,), dtype=_f32, device=dev)
        ws = _lib.workspace(dev, 512 * N)
        rc = lib.dss_renderable_mean_clamp(_lib.ptr(values), _lib.ptr(world), _lib.ptr(V), _lib.ptr(znear), _lib.ptr(zfar),
                                           _lib.ptr(first), _lib.ptr(num), N, int(shared_cloud), float(scale), float(lo),
                                           float(hi), float(fallback), int(min_points), Pw if per_cam else 0, _lib.ptr(out),
                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_renderable_mean_clamp")
    return out


def knn_kth_sqdist_view(points, cloud_to_packed_first_idx, num_points_per_cloud, K: int, V, znear, zfar, shared_cloud: bool,
                        radius=None):
    """`knn_kth_sqdist` in the reference's order under depth culling (``dss_knn_kth_sqdist_view``): every camera drops the
    points outside its [znear, zfar] BEFORE the neighbour search (rasterizer.py:599, 183-217, 310-326).  ``shared_cloud``:
    one cloud, N cameras -> (N, P) (row c: among the points camera c keeps; 0 for the ones it drops); else cloud n belongs to
    camera n -> (P,)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    dev = points.device
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    V = _lib.require_gpu(V, "V", _f32)
    znear = _lib.require_gpu(znear, "znear", _f32)
    zfar = _lib.require_gpu(zfar, "zfar", _f32)
    N, P, n_cams = first.shape[0], points.shape[0], V.shape[0]
    with torch.cuda.device(dev):
        out = torch.empty((n_cams, P) if shared_cloud else (P,), dtype=_f32, device=dev)
        ws = _lib.workspace(dev, lib.dss_knn_workspace(N, P))
        rc = lib.dss_knn_kth_sqdist_view(_lib.ptr(points), _lib.ptr(first), _lib.ptr(num), N, P, int(K),
                                         float(radius) if radius is not None else -1.0, _lib.ptr(V), _lib.ptr(znear),
                                         _lib.ptr(zfar), n_cams, int(shared_cloud), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(dev))
    _lib.check(rc, "dss_knn_kth_sqdist_view")
    return out


def _phong_common(world, normals, rgb, first, num, shared_cloud, ambient, diffuse_color, specular_color, light_vec,
                  cam_center):
    world = _lib.require_gpu(world, "world", _f32)
    normals = _lib.require_gpu(normals, "normals", _f32)
    rgb = _lib.require_gpu(rgb, "rgb", _f32)
    first = _lib.require_gpu(first, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num, "num_points_per_cloud", _i64)
    N, Pw = first.shape[0], world.shape[0]
    P = N * Pw if shared_cloud else Pw
    amb = _lib.require_gpu(ambient, "ambient", _f32)
    kd = _lib.require_gpu(diffuse_color, "diffuse_color", _f32)
    ks = _lib.require_gpu(specular_color, "specular_color", _f32)
    lv = _lib.require_gpu(light_vec, "light_vec", _f32)
    cam = _lib.require_gpu(cam_center, "cam_center", _f32)
    L = kd.shape[1] if kd.dim() == 3 else 0
    if tuple(rgb.shape) != (P, 3) or normals.shape != world.shape or tuple(amb.shape) != (N, 3) \
            or tuple(kd.shape) != (N, L, 3) or tuple(ks.shape) != (N, L, 3) or tuple(lv.shape) != (N, L, 3) \
            or tuple(cam.shape) != (N, 3):
        raise RuntimeError("phong: need rgb (P,3), ambient / cam_center (N,3), light tensors (N,L,3)")
    return world, normals, rgb, first, num, N, Pw, P, amb, kd, ks, lv, cam, L


def phong_forward(world, normals, rgb, cloud_to_packed_first_idx, num_points_per_cloud, ambient, diffuse_color,
                  specular_color, light_vec, point_lights: bool, cam_center, shininess: float = 64.0,
                  shared_cloud: bool = False):
    """Phong shading of the points (LightingTexture.forward, texture.py:65-125; lighting.py:10-172) -> (P,3)."""
    lib = _lib.load()
    world, normals, rgb, first, num, N, Pw, P, amb, kd, ks, lv, cam, L = _phong_common(
        world, normals, rgb, cloud_to_packed_first_idx, num_points_per_cloud, shared_cloud, ambient, diffuse_color,
        specular_color, light_vec, cam_center)
    dev = world.device
    with torch.cuda.device(dev):
        out = torch.empty((P, 3), dtype=_f32, device=dev)
        rc = lib.dss_phong_forward(_lib.ptr(world), _lib.ptr(normals), _lib.ptr(rgb), _lib.ptr(first), _lib.ptr(num), N, Pw,
                                   int(shared_cloud), _lib.ptr(amb), _lib.ptr(kd), _lib.ptr(ks), _lib.ptr(lv), L,
                                   int(point_lights), _lib.ptr(cam), float(shininess), _lib.ptr(out),
                                   _lib.stream_ptr(dev))
    _lib.check(rc, "dss_phong_forward")
    return out


def phong_backward(grad_out, world, normals, rgb, cloud_to_packed_first_idx, num_points_per_cloud, ambient,
                   diffuse_color, specular_color, light_vec, point_lights: bool, cam_center, shininess: float = 64.0,
                   shared_cloud: bool = False):
    """-> (grad_world (Pw,3), grad_normals (Pw,3), grad_rgb (P,3))."""
    lib = _lib.load()
    world, normals, rgb, first, num, N, Pw, P, amb, kd, ks, lv, cam, L = _phong_common(
        world, normals, rgb, cloud_to_packed_first_idx, num_points_per_cloud, shared_cloud, ambient, diffuse_color,
        specular_color, light_vec, cam_center)
    grad_out = _lib.require_gpu(grad_out, "grad_out", _f32)
    if tuple(grad_out.shape) != (P, 3):
        raise RuntimeError("phong_backward: grad_out must be (P,3)")
    dev = world.device
    with torch.cuda.device(dev):
        gw = torch.empty((Pw, 3), dtype=_f32, device=dev)
        gn = torch.empty((Pw, 3), dtype=_f32, device=dev)
        gc = torch.empty((P, 3), dtype=_f32, device=dev)
        rc = lib.dss_phong_backward(_lib.ptr(grad_out), _lib.ptr(world), _lib.ptr(normals), _lib.ptr(rgb), _lib.ptr(first),
                                    _lib.ptr(num), N, Pw, int(shared_cloud), _lib.ptr(amb), _lib.ptr(kd), _lib.ptr(ks),
                                    _lib.ptr(lv), L, int(point_lights), _lib.ptr(cam), float(shininess), _lib.ptr(gw),
                                    _lib.ptr(gn), _lib.ptr(gc), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_phong_backward")
    return gw, gn, gc


def _mask_u8(mask, name, P, dev):
    if mask is None:
        return None
    mask = _lib.require_gpu(mask, name)
    if mask.numel() != P:
        raise RuntimeError("dss_amd: %s must have one entry per packed point (%d), got %d" % (name, P, mask.numel()))
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)   # same bytes: no conversion kernel
    return mask.to(torch.uint8).contiguous()


def _knn_lists(knn_dists, knn_idx, P, need_dists=True):
    knn_idx = _lib.require_gpu(knn_idx, "knn_idx", _i64)
    if knn_idx.dim() != 2 or knn_idx.shape[0] != P:
        raise RuntimeError("dss_amd: knn_idx must be (P, K) with P = %d, got %s" % (P, tuple(knn_idx.shape)))
    if need_dists:
        knn_dists = _lib.require_gpu(knn_dists, "knn_dists", _f32)
        if knn_dists.shape != knn_idx.shape:
            raise RuntimeError("dss_amd: knn_dists %s and knn_idx %s differ" % (tuple(knn_dists.shape), tuple(knn_idx.shape)))
    return knn_dists, knn_idx, int(knn_idx.shape[1])


def mollify_normals(normals, knn_dists, knn_idx, keep, cloud_to_packed_first_idx, num_points_per_cloud):
    """Robust normal mollification of the regularisers (SurfaceLoss._denoise_normals with get_phi weights,
    losses.py:181-222, 262-278) on the packed neighbour lists of :func:`knn_points` -> (P,3).  ``keep`` (P,) bool marks
    the points whose own normal is kept (visibility & inmask); None mollifies every point."""
    lib = _lib.load()
    normals = _lib.require_gpu(normals, "normals", _f32)
    dev, P = normals.device, normals.shape[0]
    knn_dists, knn_idx, K = _knn_lists(knn_dists, knn_idx, P)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    keep = _mask_u8(keep, "keep", P, dev)
    with torch.cuda.device(dev):
        out = torch.empty_like(normals)
        rc = lib.dss_mollify_normals(_lib.ptr(normals), _lib.ptr(knn_dists), _lib.ptr(knn_idx), _lib.ptr(keep),
                                     _lib.ptr(first), _lib.ptr(num), first.shape[0], P, K, _lib.ptr(out),
                                     _lib.stream_ptr(dev))
    _lib.check(rc, "dss_mollify_normals")
    return out


def projection_loss(points, mollified, knn_dists, knn_idx, visible, cloud_to_packed_first_idx, num_points_per_cloud,
                    sharpness_sigma: float, grad_loss=None, want_loss: bool = True, want_grad: bool = False):
    """ProjectionLoss.compute (losses.py:296-392) per packed point -> (loss (P,) or None, grad_points (P,3) or None);
    grad_points = d loss_i / d p_i * grad_loss_i (grad_loss None = ones)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    mollified = _lib.require_gpu(mollified, "mollified", _f32)
    dev, P = points.device, points.shape[0]
    knn_dists, knn_idx, K = _knn_lists(knn_dists, knn_idx, P)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    visible = _mask_u8(visible, "visible", P, dev)
    if grad_loss is not None:
        grad_loss = _lib.require_gpu(grad_loss, "grad_loss", _f32)
        if grad_loss.numel() != P:
            raise RuntimeError("dss_amd: grad_loss must be (P,)")
    with torch.cuda.device(dev):
        loss = torch.empty((P,), dtype=_f32, device=dev) if want_loss else None
        grad = torch.empty((P, 3), dtype=_f32, device=dev) if want_grad else None
        rc = lib.dss_projection_loss(_lib.ptr(points), _lib.ptr(mollified), _lib.ptr(knn_dists), _lib.ptr(knn_idx),
                                     _lib.ptr(visible), _lib.ptr(first), _lib.ptr(num), first.shape[0], P, K,
                                     float(sharpness_sigma), _lib.ptr(grad_loss), _lib.ptr(loss), _lib.ptr(grad),
                                     _lib.stream_ptr(dev))
    _lib.check(rc, "dss_projection_loss")
    return loss, grad


def repulsion_loss(points, mollified, knn_idx, cloud_to_packed_first_idx, num_points_per_cloud, sharpness_sigma: float,
                   filter_scale: float, grad_loss=None, want_loss: bool = True, want_grad: bool = False):
    """RepulsionLoss.compute (losses.py:395-492) per packed point -> (loss (P,3) or None, grad_points (P,3) or None)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    mollified = _lib.require_gpu(mollified, "mollified", _f32)
    dev, P = points.device, points.shape[0]
    _, knn_idx, K = _knn_lists(None, knn_idx, P, need_dists=False)
    first = _lib.require_gpu(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", _i64)
    num = _lib.require_gpu(num_points_per_cloud, "num_points_per_cloud", _i64)
    if grad_loss is not None:
        grad_loss = _lib.require_gpu(grad_loss, "grad_loss", _f32)
        if grad_loss.numel() != 3 * P:
            raise RuntimeError("dss_amd: grad_loss must be (P,3)")
    N = first.shape[0]
    with torch.cuda.device(dev):
        loss = torch.empty((P, 3), dtype=_f32, device=dev) if want_loss else None
        grad = torch.empty((P, 3), dtype=_f32, device=dev) if want_grad else None
        ws = _lib.workspace(dev, 24 * N)
        rc = lib.dss_repulsion_loss(_lib.ptr(points), _lib.ptr(mollified), _lib.ptr(knn_idx), _lib.ptr(first), _lib.ptr(num),
                                    N, P, K, float(sharpness_sigma), float(filter_scale), _lib.ptr(grad_loss), _lib.ptr(loss),
                                    _lib.ptr(grad), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_repulsion_loss")
    return loss, grad


def _image_loss_args(rgba, target_rgb, target_mask):
    rgba = _lib.require_gpu(rgba, "rgba", _f32)
    if rgba.dim() != 4 or rgba.shape[-1] != 4:
        raise RuntimeError("dss_amd: rgba must be (N,H,W,4), got %s" % (tuple(rgba.shape),))
    N, H, W, _ = rgba.shape
    if not isinstance(target_rgb, torch.Tensor) or not target_rgb.is_cuda or target_rgb.dtype != _f32:
        raise RuntimeError("dss_amd: target_rgb must be a float32 GPU tensor (no CPU fallback)")
    if tuple(target_rgb.shape) != (N, H, W, 3):
        raise RuntimeError("dss_amd: target_rgb must be (N,H,W,3) = %s (a permuted NCHW view is fine), got %s"
                           % ((N, H, W, 3), tuple(target_rgb.shape)))
    target_mask = _lib.require_gpu(target_mask, "target_mask", _f32).reshape(N, H, W)
    return rgba, target_rgb, target_mask, N, H, W


def image_loss_forward(rgba, target_rgb, target_mask, lambda_rgb: float, lambda_silhouette: float):
    """Trainer.calc_dr_loss (trainer.py:332-372) on the rendered (N,H,W,4) image -> (losses (4,) = total, weighted rgb
    term, weighted silhouette term, IoU term; sums (N+1,5) float64 for :func:`image_loss_backward`).  ``target_rgb``
    (N,H,W,3) may be any strided view (e.g. ``img.permute(0, 2, 3, 1)``); no host synchronisation."""
    lib = _lib.load()
    rgba, target_rgb, target_mask, N, H, W = _image_loss_args(rgba, target_rgb, target_mask)
    dev = rgba.device
    sn, sh, sw, sc = target_rgb.stride()
    with torch.cuda.device(dev):
        losses = torch.empty((4,), dtype=_f32, device=dev)
        sums = torch.empty((N + 1, 5), dtype=torch.float64, device=dev)
        ws = _lib.workspace(dev, lib.dss_image_loss_workspace(N, H, W))
        rc = lib.dss_image_loss_forward(_lib.ptr(rgba), _lib.ptr(target_rgb), sn, sh, sw, sc, _lib.ptr(target_mask), N, H, W,
                                        float(lambda_rgb), float(lambda_silhouette), _lib.ptr(sums), _lib.ptr(losses),
                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_image_loss_forward")
    return losses, sums


def image_loss_backward(rgba, target_rgb, target_mask, lambda_rgb: float, lambda_silhouette: float, sums, grad_total=None):
    """Gradient of the total image loss w.r.t. the rendered image, (N,H,W,4), scaled by the device scalar
    ``grad_total`` (None = 1)."""
    lib = _lib.load()
    rgba, target_rgb, target_mask, N, H, W = _image_loss_args(rgba, target_rgb, target_mask)
    dev = rgba.device
    sums = _lib.require_gpu(sums, "sums", torch.float64)
    if tuple(sums.shape) != (N + 1, 5):
        raise RuntimeError("dss_amd: sums must be (N+1,5) from image_loss_forward")
    if grad_total is not None:
        grad_total = _lib.require_gpu(grad_total, "grad_total", _f32).reshape(1)
    sn, sh, sw, sc = target_rgb.stride()
    with torch.cuda.device(dev):
        grad = torch.empty_like(rgba)
        rc = lib.dss_image_loss_backward(_lib.ptr(rgba), _lib.ptr(target_rgb), sn, sh, sw, sc, _lib.ptr(target_mask), N, H, W,
                                         float(lambda_rgb), float(lambda_silhouette), _lib.ptr(sums), _lib.ptr(grad_total),
                                         _lib.ptr(grad), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_image_loss_backward")
    return grad


def points_inmask(points, M, mask_img, visible=None):
    """In-mask filter of the regularisers (point_modeling.py:183-208): bool (P,) = visible & any over the views of
    (target mask bilinearly sampled at the point's projection != 0).  ``points`` (P,3) world positions of one cloud,
    ``M`` (N,4,4) full projection matrices, ``mask_img`` (N,H,W) or (N,1,H,W)."""
    lib = _lib.load()
    points = _lib.require_gpu(points, "points", _f32)
    M = _lib.require_gpu(M, "M", _f32)
    dev, P, N = points.device, points.shape[0], M.shape[0]
    if not isinstance(mask_img, torch.Tensor) or not mask_img.is_cuda:
        raise RuntimeError("dss_amd: mask_img must be a GPU tensor (no CPU fallback)")
    if mask_img.dim() == 4:
        mask_img = mask_img[:, 0]
    if mask_img.dim() != 3 or mask_img.shape[0] != N:
        raise RuntimeError("dss_amd: mask_img must be (N,H,W) or (N,1,H,W) with N = %d views, got %s" % (N, tuple(mask_img.shape)))
    mask_img = mask_img.to(_f32).contiguous()
    visible = _mask_u8(visible, "visible", P, dev)
    with torch.cuda.device(dev):
        out = torch.empty((P,), dtype=torch.uint8, device=dev)
        rc = lib.dss_points_inmask(_lib.ptr(points), _lib.ptr(M), _lib.ptr(mask_img), _lib.ptr(visible), N, P,
                                   mask_img.shape[1], mask_img.shape[2], _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_points_inmask")
    return out.view(torch.bool)   # the kernel writes 0 / 1


_band_row_cache = {}


def _band_row_index(row0, row1, cyc, device):
    """image rows of the tile-row-cyclic band (row0, row1, cyc), in band order, as a cached int64 device tensor"""
    key = (row0, row1, cyc, device)
    t = _band_row_cache.get(key)
    if t is None:
        if len(_band_row_cache) > 64:
            _band_row_cache.clear()
        rows = [r0 + i for r0 in range(row0, row1, 8 * cyc) for i in range(8) if r0 + i < row1]
        t = _band_row_cache[key] = torch.tensor(rows, dtype=_i64, device=device)
    return t


def band_targets(target_rgb, target_mask, rows):
    """The rows ``rows`` of the targets of a band loss, gathered ONCE (a tile-row-cyclic band costs two index_select launches
    per call otherwise): pass the result as ``band_targets=`` to `image_loss_band_sums` / `image_loss_band_backward` while the
    targets stay the same."""
    row0, row1, cyc = _band(rows, None) if len(rows) > 2 else (int(rows[0]), int(rows[1]), 1)
    N, H, W = target_rgb.shape[0], target_rgb.shape[1], target_rgb.shape[2]
    target_mask = target_mask.to(_f32).reshape(N, H, W)
    if cyc > 1:
        ri = _band_row_index(row0, row1, cyc, target_rgb.device)
        return target_rgb.index_select(1, ri).contiguous(), target_mask.index_select(1, ri).contiguous()
    return target_rgb[:, row0:row1], target_mask[:, row0:row1]


def _band_args(rgba_band, target_rgb, target_mask, rows, band_targets=None, keep_strides=False):
    """Row band of an image loss: the band render (N,rows,W,4) against the FULL targets (N,H,W,3) / (N,H,W); returns the
    tensors plus the targets' band and the image stride of its mask.  ``rows`` = (row0, row1) for a contiguous band (views of
    the targets, no copy) or (row0, row1, cycle) for a tile-row-cyclic one (`RowPartition(cyclic=True).rows`: the owned
    rows of the targets are gathered once per call).  ``keep_strides``: the band has been vetted by `_strided_band` and is
    taken as it is (not made contiguous)."""
    if not keep_strides:
        rgba_band = _lib.require_gpu(rgba_band, "rgba_band", _f32)
    row0, row1, cyc = _band(rows, None) if rows is not None and len(rows) > 2 else (int(rows[0]), int(rows[1]), 1)
    want = band_rows(row0, row1, cyc)
    if rgba_band.dim() != 4 or rgba_band.shape[-1] != 4 or rgba_band.shape[1] != want:
        raise RuntimeError("dss_amd: rgba_band must be (N, %d, W, 4) for rows %s, got %s" % (want, tuple(rows), tuple(rgba_band.shape)))
    N, nr, W, _ = rgba_band.shape
    if not isinstance(target_rgb, torch.Tensor) or not target_rgb.is_cuda or target_rgb.dtype != _f32:
        raise RuntimeError("dss_amd: target_rgb must be a float32 GPU tensor (no CPU fallback)")
    H = target_rgb.shape[1]
    if target_rgb.dim() != 4 or tuple(target_rgb.shape) != (N, H, W, 3) or not (0 <= row0 <= row1 <= H):
        raise RuntimeError("dss_amd: target_rgb must be the full (N,H,W,3) target with 0 <= row0 <= row1 <= H")
    target_mask = _lib.require_gpu(target_mask, "target_mask", _f32).reshape(N, H, W)
    if band_targets is not None:
        band_rgb, band_mask = band_targets
        if tuple(band_rgb.shape) != (N, nr, W, 3) or tuple(band_mask.shape) != (N, nr, W):
            raise RuntimeError("dss_amd: band_targets must be (N,%d,W,3) and (N,%d,W)" % (nr, nr))
        return rgba_band, band_rgb, band_mask, target_mask, N, nr, W, H, int(band_mask.stride(0))
    if cyc > 1:
        ri = _band_row_index(row0, row1, cyc, rgba_band.device)
        band_rgb = target_rgb.index_select(1, ri)
        band_mask = target_mask.index_select(1, ri)
        return rgba_band, band_rgb, band_mask, target_mask, N, nr, W, H, nr * W
    band_rgb = target_rgb[:, row0:row1]
    band_mask = target_mask[:, row0:row1]
    return rgba_band, band_rgb, band_mask, target_mask, N, nr, W, H, H * W


def image_loss_band_sums(rgba_band, target_rgb, target_mask, rows, band_targets=None):
    """Per-image sums of ``Trainer.calc_dr_loss`` over the row band ``rows = (row0, row1)`` -> float64 (N+1,5) whose
    first N rows are filled; all-reduce (SUM) ``sums[:N]`` over the ranks, then :func:`image_loss_from_sums`."""
    lib = _lib.load()
    rgba_band, band_rgb, band_mask, _keep, N, nr, W, H, mstride = _band_args(rgba_band, target_rgb, target_mask, rows, band_targets)
    dev = rgba_band.device
    with torch.cuda.device(dev):
        sums = torch.zeros((N + 1, 5), dtype=torch.float64, device=dev)
        if nr > 0:
            sn, sh, sw, sc = band_rgb.stride()
            ws = _lib.workspace(dev, lib.dss_image_loss_workspace(N, nr, W))
            rc = lib.dss_image_loss_band_sums(_lib.ptr(rgba_band), _lib.ptr(band_rgb), sn, sh, sw, sc,
                                              _lib.ptr(band_mask), mstride, N, nr, W, _lib.ptr(sums),
                                              _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
            _lib.check(rc, "dss_image_loss_band_sums")
    return sums


def image_loss_from_sums(sums, image_size, lambda_rgb: float, lambda_silhouette: float):
    """Totals row + losses (4,) from the (all-reduced) per-image sums; ``image_size = (H, W)`` of the FULL image."""
    lib = _lib.load()
    sums = _lib.require_gpu(sums, "sums", torch.float64)
    N = sums.shape[0] - 1
    dev = sums.device
    with torch.cuda.device(dev):
        losses = torch.empty((4,), dtype=_f32, device=dev)
        rc = lib.dss_image_loss_from_sums(_lib.ptr(sums), N, int(image_size[0]), int(image_size[1]), float(lambda_rgb),
                                          float(lambda_silhouette), _lib.ptr(losses), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_image_loss_from_sums")
    return losses


def image_loss_band_backward(rgba_band, target_rgb, target_mask, rows, lambda_rgb: float, lambda_silhouette: float, sums,
                             grad_total=None, band_targets=None):
    """The band ``rows`` of d total / d rgba, (N,rows,W,4), from the reduced ``sums`` (after image_loss_from_sums)."""
    lib = _lib.load()
    rgba_band, band_rgb, band_mask, _keep, N, nr, W, H, mstride = _band_args(rgba_band, target_rgb, target_mask, rows, band_targets)
    dev = rgba_band.device
    sums = _lib.require_gpu(sums, "sums", torch.float64)
    if grad_total is not None:
        grad_total = _lib.require_gpu(grad_total, "grad_total", _f32).reshape(1)
    with torch.cuda.device(dev):
        grad = torch.empty_like(rgba_band)
        if nr > 0:
            sn, sh, sw, sc = band_rgb.stride()
            rc = lib.dss_image_loss_band_backward(_lib.ptr(rgba_band), _lib.ptr(band_rgb), sn, sh, sw, sc,
                                                  _lib.ptr(band_mask), mstride, N, nr, W, H,
                                                  float(lambda_rgb), float(lambda_silhouette), _lib.ptr(sums),
                                                  _lib.ptr(grad_total), _lib.ptr(grad), _lib.stream_ptr(dev))
            _lib.check(rc, "dss_image_loss_band_backward")
    return grad


def _strided_band(rgba_band):
    """-> (tensor, camera stride, row stride) of an (N, rows, W, 4) float32 band whose pixels and rows of pixels are contiguous
    (the (row, camera, col, channel) send buffer of the multi-GPU renderer through its (N, rows, W, 4) view); (t, 0, 0) for a
    dense band; anything else is copied"""
    if not isinstance(rgba_band, torch.Tensor) or not rgba_band.is_cuda or rgba_band.dtype != _f32:
        raise RuntimeError("dss_amd: rgba_band must be a float32 GPU tensor (no CPU fallback)")
    if rgba_band.dim() != 4 or rgba_band.shape[-1] != 4:
        raise RuntimeError("dss_amd: rgba_band must be (N,rows,W,4), got %s" % (tuple(rgba_band.shape),))
    if rgba_band.is_contiguous():
        return rgba_band, 0, 0
    sn, sh, sw, sc = rgba_band.stride()
    if sc == 1 and sw == 4 and sn % 4 == 0 and sh % 4 == 0 and sh >= 4 * rgba_band.shape[2] and rgba_band.data_ptr() % 16 == 0:
        return rgba_band, int(sn), int(sh)
    return rgba_band.contiguous(), 0, 0


def image_loss_band_partials(rgba_band, target_rgb, target_mask, rows, band_targets=None, out=None):
    """First launch of the two-launch band loss (``dss_image_loss_band_partials``): the block partials of the band's five
    per-image sums -> float64 (N, 64, 5).  All-reduce THEM (SUM) over the ranks, then `image_loss_band_backward_partials`.
    ``rgba_band`` may be the strided (N, rows, W, 4) view of a (row, camera, col, channel) buffer (no copy).
    ``out``: a caller-owned buffer of that shape (a step that replays as a graph all-reduces a static one)."""
    lib = _lib.load()
    view, rsn, rsh = _strided_band(rgba_band)
    _c, band_rgb, band_mask, _keep, N, nr, W, H, mstride = _band_args(view, target_rgb, target_mask, rows, band_targets,
                                                                      keep_strides=True)
    dev = view.device
    with torch.cuda.device(dev):
        n = lib.dss_image_loss_band_partials_count(N)
        part = torch.empty((N, n // (5 * N), 5), dtype=torch.float64, device=dev) if out is None else out
        if part.numel() != n or part.dtype != torch.float64 or not part.is_contiguous():
            raise RuntimeError("dss_amd: out must be a contiguous float64 tensor of %d elements" % n)
        sn, sh, sw, sc = band_rgb.stride() if nr > 0 else (0, 0, 0, 0)
        rc = lib.dss_image_loss_band_partials(_lib.ptr(view), _lib.ptr(band_rgb), sn, sh, sw, sc, _lib.ptr(band_mask), mstride,
                                              N, nr, W, rsn, rsh, _lib.ptr(part), _lib.stream_ptr(dev))
    _lib.check(rc, "dss_image_loss_band_partials")
    return part


def image_loss_band_backward_partials(rgba_band, target_rgb, target_mask, rows, lambda_rgb: float, lambda_silhouette: float,
                                      partials, grad_total=None, band_targets=None, want_sums: bool = False, alpha_out=None):
    """Second launch: from the ALL-REDUCED partials -> (band of d total / d rgba (N,rows,W,4) dense, losses (4,) = total,
    weighted rgb term, weighted silhouette term, IoU term[, sums (N+1,5) float64]) -- identical bits on every rank.
    ``alpha_out`` (N, rows, W) float32, last dimension contiguous (any camera / row strides): receives the alpha channel of
    the gradient a second time -- the send buffer of the owner form's alpha-plane exchange, no extraction copy."""
    lib = _lib.load()
    view, rsn, rsh = _strided_band(rgba_band)
    _c, band_rgb, band_mask, _keep, N, nr, W, H, mstride = _band_args(view, target_rgb, target_mask, rows, band_targets,
                                                                      keep_strides=True)
    dev = view.device
    partials = _lib.require_gpu(partials, "partials", torch.float64)
    if partials.numel() != lib.dss_image_loss_band_partials_count(N):
        raise RuntimeError("dss_amd: partials must come from image_loss_band_partials")
    if grad_total is not None:
        grad_total = _lib.require_gpu(grad_total, "grad_total", _f32).reshape(1)
    a_p, asn, ash = None, 0, 0
    if alpha_out is not None:
        if tuple(alpha_out.shape) != (N, nr, W) or alpha_out.dtype != _f32 or not alpha_out.is_cuda or (nr > 0 and alpha_out.stride(2) != 1):
            raise RuntimeError("dss_amd: alpha_out must be a float32 GPU tensor (N,%d,%d) with contiguous rows" % (nr, W))
        a_p, asn, ash = _lib.ptr(alpha_out), int(alpha_out.stride(0)), int(alpha_out.stride(1))
    with torch.cuda.device(dev):
        grad = torch.empty((N, nr, W, 4), dtype=_f32, device=dev)
        losses = torch.empty((4,), dtype=_f32, device=dev)
        sums = torch.empty((N + 1, 5), dtype=torch.float64, device=dev) if want_sums else None
        sn, sh, sw, sc = band_rgb.stride() if nr > 0 else (0, 0, 0, 0)
        rc = lib.dss_image_loss_band_backward_partials(_lib.ptr(view), _lib.ptr(band_rgb), sn, sh, sw, sc, _lib.ptr(band_mask),
                                                       mstride, N, nr, W, H, float(lambda_rgb), float(lambda_silhouette),
                                                       _lib.ptr(partials), _lib.ptr(grad_total), _lib.ptr(grad), _lib.ptr(losses),
                                                       _lib.ptr(sums), rsn, rsh, a_p, asn, ash, _lib.stream_ptr(dev))
    _lib.check(rc, "dss_image_loss_band_backward_partials")
    return (grad, losses, sums) if want_sums else (grad, losses)
