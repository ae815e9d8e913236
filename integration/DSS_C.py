"""`DSS/_C.py` -- drop this file into the reference checkout in place of the pybind11 module that `setup.py` builds from
`DSS/csrc/ext.cpp` (ext.cpp:5-18): `from .. import _C` (rasterizer.py:21) then resolves to the MI355X library through its
C ABI (include/dss_hip.h).  Plain ctypes + torch, no dependency on the `dss_amd` Python package.  This is the binding that
INTEGRATION.md section 3 prints; `tests/test_gpu_raster.py::test_integration_stub_file_matches_the_python_mirror` runs it.

    DSS_HIP_LIBRARY=/path/to/libdss_hip.so     (default: the in-tree build next to this repository's dss_amd package)
"""
import os

import ctypes, torch
_lib = ctypes.CDLL(os.environ.get("DSS_HIP_LIBRARY") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                      "dss_amd", "csrc", "libdss_hip.so"))
_p = lambda t: ctypes.c_void_p(t.data_ptr())


def _t(t, name, dtype=torch.float32):
    """contiguous GPU tensor of the expected dtype.  The RESULT must be bound to a local that outlives the library call:
    a `.contiguous()` temporary freed before the kernel is enqueued hands its block to the next temporary."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise RuntimeError("DSS._C: %s must be a CUDA/HIP tensor (no CPU path)" % name)
    if t.dtype != dtype:
        raise RuntimeError("DSS._C: %s must be %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


def _stream(dev):   # the current stream OF THE TENSORS' DEVICE (call under `with torch.cuda.device(dev)`)
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def splat_points(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                 num_points_per_cloud, depth_merging_thres, image_size, points_per_pixel,
                 bin_size, max_points_per_bin):            # signature of ext.cpp:8 / rasterize_points.h:461
    N, P, S, K = cloud_to_packed_first_idx.shape[0], points.shape[0], image_size, points_per_pixel
    pts, ell, cut, rad = _t(points, "points"), _t(ellipse_params, "ellipse_params"), _t(cutoff_thres, "cutoff_thres"), _t(radii, "radii")
    first, num = _t(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", torch.int64), _t(num_points_per_cloud, "num_points_per_cloud", torch.int64)
    dev = pts.device
    with torch.cuda.device(dev):
        idx = torch.empty((N, S, S, K), dtype=torch.int32, device=dev)
        zbuf, qv = torch.empty((N, S, S, K), device=dev), torch.empty((N, S, S, K), device=dev)
        occ = torch.empty((N, S, S), device=dev)
        _lib.dss_splat_forward_workspace.restype = ctypes.c_size_t
        nbytes = _lib.dss_splat_forward_workspace(N, ctypes.c_int64(P), S, K, 1 if bin_size is None else bin_size)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _check(_lib.dss_splat_forward(_p(pts), _p(ell), _p(cut), _p(rad), _p(first), _p(num), N, ctypes.c_int64(P),
                                      ctypes.c_float(depth_merging_thres), S, K, 1 if bin_size is None else bin_size, 0, S,
                                      _p(idx), _p(zbuf), _p(qv), _p(occ), None, _p(ws), ctypes.c_size_t(nbytes), _stream(dev)))
    return idx, zbuf, qv, occ

def _splat_points_occ_fast_cuda_backward(points_sorted, radii_sorted, rs, grad_occ, num_points_per_cloud,
                                         cloud_to_packed_first_idx, points_grid_off, grid_params):
    # ext.cpp:14 / rasterize_points_backward.cu:227.  Called by rasterizer.py:951-952 with the VISIBLE points only;
    # the FRNN grid arguments only accelerate the reference's pixel-centric search and are not needed by the gather.
    pts, rad, rs_, go = _t(points_sorted, "points_sorted"), _t(radii_sorted, "radii_sorted"), _t(rs, "rs"), _t(grad_occ, "grad_occ")
    first, num = _t(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", torch.int64), _t(num_points_per_cloud, "num_points_per_cloud", torch.int64)
    P, (N, H, W), dev = pts.shape[0], go.shape, pts.device
    with torch.cuda.device(dev):
        every = torch.ones(P, dtype=torch.uint8, device=dev)
        grad = torch.empty((P, 3), device=dev)
        _check(_lib.dss_occ_backward(_p(pts), _p(rad), _p(every), _p(rs_), _p(go), _p(first), _p(num), N, ctypes.c_int64(P), H,
                                     0, H, 1, ctypes.c_float(-1.0), _p(grad), _stream(dev)))
    return grad[:, :2].contiguous()          # (P,2) in the order of points_sorted, like the reference

def _backward_zbuf(idx, grad_zbuf, point_z_grad):            # ext.cpp:17 / rasterize_points.h:388: in place, (P,1)
    idx_, gz = _t(idx, "idx", torch.int32), _t(grad_zbuf, "grad_zbuf")
    N, H, W, K = idx_.shape
    with torch.cuda.device(idx_.device):
        tmp = torch.zeros((point_z_grad.shape[0], 3), device=idx_.device)   # the C entry point updates column 2 of (P,3)
        _check(_lib.dss_zbuf_backward(_p(idx_), _p(gz), N, H, W, K, _p(tmp), _stream(idx_.device)))
    point_z_grad += tmp[:, 2:3]


def _check(rc):
    if rc:
        _lib.dss_last_error.restype = ctypes.c_char_p
        raise RuntimeError(_lib.dss_last_error().decode())   # reference: TORCH_CHECK -> RuntimeError


def _splat_points_naive(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx, num_points_per_cloud,
                        depth_merging_thres, image_size, points_per_pixel):            # ext.cpp:9 / rasterize_points.h:86
    return splat_points(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx, num_points_per_cloud,
                        depth_merging_thres, image_size, points_per_pixel, 0, 0)


def _rasterize_coarse(points, radii, cloud_to_packed_first_idx, num_points_per_cloud, image_size, bin_size,
                      max_points_per_bin):                                             # ext.cpp:10 / rasterize_points.h:176
    # returns the library's OPAQUE tile-list workspace (uint8) instead of the dense (N,B,B,M) table: only
    # _rasterize_fine consumes it.  The cloud ranges ride along as attributes (the fine pass needs them).
    pts, rad = _t(points, "points"), _t(radii, "radii")
    first, num = _t(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", torch.int64), _t(num_points_per_cloud, "num_points_per_cloud", torch.int64)
    N, P, S = first.shape[0], pts.shape[0], image_size
    with torch.cuda.device(pts.device):
        _lib.dss_splat_forward_workspace.restype = ctypes.c_size_t
        nbytes = _lib.dss_splat_forward_workspace(N, ctypes.c_int64(P), S, 1, 1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
        _check(_lib.dss_splat_bin(_p(pts), _p(rad), _p(first), _p(num), N, ctypes.c_int64(P), S, 0, S, _p(ws),
                                  ctypes.c_size_t(nbytes), _stream(pts.device)))
    ws._dss_ranges = (first, num)
    return ws


def _rasterize_fine(points, ellipse_params, cutoff_thres, radii, bin_points, depth_merging_thres, image_size, bin_size,
                    points_per_pixel):                                                 # ext.cpp:11 / rasterize_points.h:268
    first, num = bin_points._dss_ranges
    pts, ell, cut, rad = _t(points, "points"), _t(ellipse_params, "ellipse_params"), _t(cutoff_thres, "cutoff_thres"), _t(radii, "radii")
    N, P, S, K, dev = first.shape[0], pts.shape[0], image_size, points_per_pixel, pts.device
    with torch.cuda.device(dev):
        idx = torch.empty((N, S, S, K), dtype=torch.int32, device=dev)
        zbuf, qv = torch.empty((N, S, S, K), device=dev), torch.empty((N, S, S, K), device=dev)
        occ = torch.empty((N, S, S), device=dev)
        _check(_lib.dss_splat_fine(_p(pts), _p(ell), _p(cut), _p(rad), _p(first), _p(num), N, ctypes.c_int64(P),
                                   ctypes.c_float(depth_merging_thres), S, K, 0, S, _p(idx), _p(zbuf), _p(qv), _p(occ), None,
                                   _p(bin_points), ctypes.c_size_t(bin_points.numel()), _stream(dev)))
    return idx, zbuf, qv, occ


def _splat_points_occ_backward(points, radii, grad_occ, cloud_to_packed_first_idx, num_points_per_cloud, radii_s,
                               depth_merging_thres):                                   # ext.cpp:12 / rasterize_points.cu:604-775
    pts, rad, go = _t(points, "points"), _t(radii, "radii"), _t(grad_occ, "grad_occ")
    first, num = _t(cloud_to_packed_first_idx, "cloud_to_packed_first_idx", torch.int64), _t(num_points_per_cloud, "num_points_per_cloud", torch.int64)
    N, S, P = go.shape[0], go.shape[1], pts.shape[0]
    with torch.cuda.device(pts.device):
        grad = torch.empty((P, 2), device=pts.device)
        _check(_lib.dss_occ_backward_box(_p(pts), _p(rad), _p(go), _p(first), _p(num), N, ctypes.c_int64(P), S,
                                         ctypes.c_float(radii_s), _p(grad), _stream(pts.device)))
    return grad
