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

def splat_points(points, ellipse_params, cutoff_thres, radii, cloud_to_packed_first_idx,
                 num_points_per_cloud, depth_merging_thres, image_size, points_per_pixel,
                 bin_size, max_points_per_bin):            # signature of ext.cpp:8 / rasterize_points.h:461
    N, P, S, K = cloud_to_packed_first_idx.shape[0], points.shape[0], image_size, points_per_pixel
    dev = points.device
    idx = torch.empty((N, S, S, K), dtype=torch.int32, device=dev)
    zbuf, qv = torch.empty((N, S, S, K), device=dev), torch.empty((N, S, S, K), device=dev)
    occ = torch.empty((N, S, S), device=dev)
    _lib.dss_splat_forward_workspace.restype = ctypes.c_size_t
    nbytes = _lib.dss_splat_forward_workspace(N, ctypes.c_int64(P), S, K, 1 if bin_size is None else bin_size)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = _lib.dss_splat_forward(_p(points.contiguous()), _p(ellipse_params.contiguous()),
            _p(cutoff_thres.contiguous()), _p(radii.contiguous()), _p(cloud_to_packed_first_idx),
            _p(num_points_per_cloud), N, ctypes.c_int64(P), ctypes.c_float(depth_merging_thres), S, K,
            1 if bin_size is None else bin_size, 0, S, _p(idx), _p(zbuf), _p(qv), _p(occ), None,
            _p(ws), ctypes.c_size_t(nbytes), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        _lib.dss_last_error.restype = ctypes.c_char_p
        raise RuntimeError(_lib.dss_last_error().decode())   # reference: TORCH_CHECK -> RuntimeError
    return idx, zbuf, qv, occ

def _splat_points_occ_fast_cuda_backward(points_sorted, radii_sorted, rs, grad_occ, num_points_per_cloud,
                                         cloud_to_packed_first_idx, points_grid_off, grid_params):
    # ext.cpp:14 / rasterize_points_backward.cu:227.  Called by rasterizer.py:951-952 with the VISIBLE points only;
    # the FRNN grid arguments only accelerate the reference's pixel-centric search and are not needed by the gather.
    P, (N, H, W), dev = points_sorted.shape[0], grad_occ.shape, points_sorted.device
    every = torch.ones(P, dtype=torch.uint8, device=dev)
    grad = torch.empty((P, 3), device=dev)
    rc = _lib.dss_occ_backward(_p(points_sorted.contiguous()), _p(radii_sorted.contiguous()), _p(every),
            _p(rs.contiguous()), _p(grad_occ.contiguous()), _p(cloud_to_packed_first_idx), _p(num_points_per_cloud),
            N, ctypes.c_int64(P), H, 0, H, 1, ctypes.c_float(-1.0), _p(grad),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(_lib.dss_last_error().decode())
    return grad[:, :2].contiguous()          # (P,2) in the order of points_sorted, like the reference

def _backward_zbuf(idx, grad_zbuf, point_z_grad):            # ext.cpp:17 / rasterize_points.h:388: in place, (P,1)
    N, H, W, K = idx.shape
    tmp = torch.zeros((point_z_grad.shape[0], 3), device=idx.device)   # the C entry point updates column 2 of (P,3)
    rc = _lib.dss_zbuf_backward(_p(idx.contiguous()), _p(grad_zbuf.contiguous()), N, H, W, K, _p(tmp),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(_lib.dss_last_error().decode())
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
    N, P, S = cloud_to_packed_first_idx.shape[0], points.shape[0], image_size
    _lib.dss_splat_forward_workspace.restype = ctypes.c_size_t
    nbytes = _lib.dss_splat_forward_workspace(N, ctypes.c_int64(P), S, 1, 1)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=points.device)
    _check(_lib.dss_splat_bin(_p(points.contiguous()), _p(radii.contiguous()), _p(cloud_to_packed_first_idx),
                              _p(num_points_per_cloud), N, ctypes.c_int64(P), S, 0, S, _p(ws), ctypes.c_size_t(nbytes),
                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ws._dss_ranges = (cloud_to_packed_first_idx, num_points_per_cloud)
    return ws


def _rasterize_fine(points, ellipse_params, cutoff_thres, radii, bin_points, depth_merging_thres, image_size, bin_size,
                    points_per_pixel):                                                 # ext.cpp:11 / rasterize_points.h:268
    first, num = bin_points._dss_ranges
    N, P, S, K, dev = first.shape[0], points.shape[0], image_size, points_per_pixel, points.device
    idx = torch.empty((N, S, S, K), dtype=torch.int32, device=dev)
    zbuf, qv = torch.empty((N, S, S, K), device=dev), torch.empty((N, S, S, K), device=dev)
    occ = torch.empty((N, S, S), device=dev)
    _check(_lib.dss_splat_fine(_p(points.contiguous()), _p(ellipse_params.contiguous()), _p(cutoff_thres.contiguous()),
                               _p(radii.contiguous()), _p(first), _p(num), N, ctypes.c_int64(P),
                               ctypes.c_float(depth_merging_thres), S, K, 0, S, _p(idx), _p(zbuf), _p(qv), _p(occ), None,
                               _p(bin_points), ctypes.c_size_t(bin_points.numel()),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return idx, zbuf, qv, occ


def _splat_points_occ_backward(points, radii, grad_occ, cloud_to_packed_first_idx, num_points_per_cloud, radii_s,
                               depth_merging_thres):                                   # ext.cpp:12 / rasterize_points.cu:604-775
    N, S, P = grad_occ.shape[0], grad_occ.shape[1], points.shape[0]
    grad = torch.empty((P, 2), device=points.device)
    _check(_lib.dss_occ_backward_box(_p(points.contiguous()), _p(radii.contiguous()), _p(grad_occ.contiguous()),
                                     _p(cloud_to_packed_first_idx), _p(num_points_per_cloud), N, ctypes.c_int64(P), S,
                                     ctypes.c_float(radii_s), _p(grad),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return grad
