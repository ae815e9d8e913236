"""ctypes binding of libdss_hip.so (the C ABI declared in include/dss_hip.h).

This module is plumbing only: it turns torch tensors into raw device pointers + sizes, passes
torch's current HIP stream, and raises ``RuntimeError`` with ``dss_last_error()`` when an entry
point returns a negative status (the reference raises RuntimeError from TORCH_CHECK / AT_ERROR,
DSS/csrc/rasterize_points.h:474-488).  There is deliberately NO CPU or pure-torch fallback: if
the shared library is missing or a tensor is not on a GPU the call fails loudly.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DSS_HIP_LIBRARY: another build of the same library -- development A/B runs; the default is the in-tree build)
LIB_PATH = os.environ.get("DSS_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libdss_hip.so")

_lib = None
_lock = threading.Lock()

_c_int, _c_i64, _c_f32, _c_vp, _c_sz = (ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                         ctypes.c_void_p, ctypes.c_size_t)

# name -> (restype, argtypes); must list every symbol of include/dss_hip.h
SIGNATURES = {
    "dss_version": (_c_int, []),
    "dss_last_error": (ctypes.c_char_p, []),
    "dss_band_rows": (_c_int, [_c_int, _c_int, _c_int]),
    "dss_set_option": (_c_int, [_c_int, _c_int]),
    "dss_get_option": (_c_int, [_c_int]),
    "dss_splat_forward_workspace": (_c_sz, [_c_int, _c_i64, _c_int, _c_int, _c_int]),
    "dss_splat_forward_clean_bytes": (_c_sz, [_c_int, _c_i64, _c_int]),
    "dss_splat_forward": (_c_int, [_c_vp] * 6 + [_c_int, _c_i64, _c_f32, _c_int, _c_int, _c_int, _c_int, _c_int]
                          + [_c_vp] * 5 + [_c_vp, _c_sz, _c_vp]),
    "dss_splat_bin": (_c_int, [_c_vp] * 4 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "dss_splat_fine": (_c_int, [_c_vp] * 6 + [_c_int, _c_i64, _c_f32, _c_int, _c_int, _c_int, _c_int]
                       + [_c_vp] * 5 + [_c_vp, _c_sz, _c_vp]),
    "dss_splat_fine_blend": (_c_int, [_c_vp] * 6 + [_c_int, _c_i64, _c_f32, _c_int, _c_int, _c_int, _c_int]
                             + [_c_vp] * 5 + [_c_vp, _c_vp, _c_int, _c_vp, _c_vp] + [_c_vp, _c_sz, _c_vp]),
    "dss_backward_radius_workspace": (_c_sz, [_c_int, _c_i64]),
    "dss_backward_radius": (_c_int, [_c_vp] * 4 + [_c_int, _c_i64, _c_f32, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dss_occ_backward_box": (_c_int, [_c_vp] * 5 + [_c_int, _c_i64, _c_int, _c_f32, _c_vp, _c_vp]),
    "dss_occ_backward": (_c_int, [_c_vp] * 7 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_f32, _c_vp, _c_vp]),
    "dss_zbuf_backward": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "dss_clip_grad": (_c_int, [_c_vp, _c_i64, _c_f32, _c_vp]),
    "dss_splat_backward_workspace": (_c_sz, [_c_int, _c_i64]),
    "dss_splat_backward": (_c_int, [_c_vp] * 8 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_f32, _c_f32,
                                                  _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dss_render_forward_workspace": (_c_sz, [_c_int, _c_i64, _c_int, _c_int]),
    "dss_render_forward": (_c_int, [_c_vp] * 12 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_f32, _c_f32, _c_f32,
                                                  _c_int, _c_int, _c_int, _c_vp, _c_int] + [_c_vp] * 12 + [_c_i64, _c_i64, _c_vp] + [_c_vp, _c_sz, _c_int, _c_vp]),
    "dss_render_backward_workspace": (_c_sz, [_c_int, _c_i64, _c_int]),
    "dss_render_backward": (_c_int, [_c_vp] * 10 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32, _c_f32]
                            + [_c_vp] * 5 + [_c_vp, _c_sz, _c_vp]),
    "dss_render_backward_owned": (_c_int, [_c_vp] * 11 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32, _c_f32]
                                  + [_c_vp] * 3 + [_c_vp, _c_sz, _c_vp]),
    "dss_render_backward_owned_plane": (_c_int, [_c_vp] * 11 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f32,
                                                               _c_f32] + [_c_vp] * 3 + [_c_vp, _c_sz, _c_vp]),
    "dss_gather_rows": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "dss_render_backward_gather": (_c_int, [_c_vp] * 10 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                                                          _c_f32, _c_f32] + [_c_vp] * 5 + [_c_vp, _c_sz, _c_vp]),
    "dss_phong_forward": (_c_int, [_c_vp] * 5 + [_c_int, _c_i64, _c_int] + [_c_vp] * 4 + [_c_int, _c_int, _c_vp, _c_f32,
                                                                                         _c_vp, _c_vp]),
    "dss_phong_backward": (_c_int, [_c_vp] * 6 + [_c_int, _c_i64, _c_int] + [_c_vp] * 4 + [_c_int, _c_int, _c_vp, _c_f32,
                                                                                          _c_vp, _c_vp, _c_vp, _c_vp]),
    "dss_mollify_normals": (_c_int, [_c_vp] * 6 + [_c_int, _c_i64, _c_int, _c_vp, _c_vp]),
    "dss_projection_loss": (_c_int, [_c_vp] * 7 + [_c_int, _c_i64, _c_int, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp]),
    "dss_repulsion_loss": (_c_int, [_c_vp] * 5 + [_c_int, _c_i64, _c_int, _c_f32, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz,
                                                  _c_vp]),
    "dss_points_inmask": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_i64, _c_int, _c_int, _c_vp, _c_vp]),
    "dss_image_loss_workspace": (_c_sz, [_c_int, _c_int, _c_int]),
    "dss_image_loss_forward": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32,
                                        _c_f32, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dss_image_loss_backward": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_int, _c_int, _c_int, _c_f32,
                                         _c_f32, _c_vp, _c_vp, _c_vp, _c_vp]),
    "dss_image_loss_band_sums": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                          _c_vp, _c_vp, _c_sz, _c_vp]),
    "dss_image_loss_from_sums": (_c_int, [_c_vp, _c_int, _c_int, _c_int, _c_f32, _c_f32, _c_vp, _c_vp]),
    "dss_image_loss_band_backward": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_int, _c_int,
                                              _c_int, _c_int, _c_f32, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp]),
    "dss_image_loss_band_partials_count": (_c_sz, [_c_int]),
    "dss_image_loss_band_partials": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_int, _c_int, _c_int,
                                              _c_i64, _c_i64, _c_vp, _c_vp]),
    "dss_image_loss_band_backward_partials": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_i64, _c_int, _c_int,
                                                       _c_int, _c_int, _c_f32, _c_f32, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                                       _c_i64, _c_i64, _c_vp, _c_i64, _c_i64, _c_vp]),
    "dss_knn_workspace": (_c_sz, [_c_int, _c_i64]),
    "dss_knn_kth_sqdist": (_c_int, [_c_vp] * 3 + [_c_int, _c_i64, _c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dss_knn_kth_sqdist_radius": (_c_int, [_c_vp] * 3 + [_c_int, _c_i64, _c_int, _c_f32, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dss_knn_points": (_c_int, [_c_vp] * 3 + [_c_int, _c_i64, _c_int, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "dss_cloud_mean_clamp": (_c_int, [_c_vp] * 3 + [_c_int, _c_f32, _c_f32, _c_f32, _c_f32, _c_int, _c_vp, _c_vp]),
    "dss_renderable_mean_clamp": (_c_int, [_c_vp] * 7 + [_c_int, _c_int, _c_f32, _c_f32, _c_f32, _c_f32, _c_int, _c_i64, _c_vp, _c_vp,
                                           _c_sz, _c_vp]),
    "dss_knn_kth_sqdist_view": (_c_int, [_c_vp] * 3 + [_c_int, _c_i64, _c_int, _c_f32, _c_vp, _c_vp, _c_vp, _c_int, _c_int, _c_vp, _c_vp,
                                         _c_sz, _c_vp]),
    "dss_blend_forward": (_c_int, [_c_vp] * 5 + [_c_int] * 5 + [_c_vp, _c_vp, _c_vp]),
    "dss_local_frames": (_c_int, [_c_vp] * 4 + [_c_int, _c_i64, _c_int, _c_vp, _c_vp, _c_vp, _c_vp]),
    "dss_point_setup": (_c_int, [_c_vp] * 12 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_f32, _c_f32]
                        + [_c_vp] * 6 + [_c_vp]),
    "dss_project_backward": (_c_int, [_c_vp] * 5 + [_c_int, _c_i64, _c_int, _c_vp, _c_vp, _c_f32, _c_vp, _c_vp]),
    "dss_project_backward_features": (_c_int, [_c_vp] * 5 + [_c_int, _c_i64, _c_int, _c_vp, _c_vp, _c_f32, _c_vp, _c_vp, _c_int,
                                                              _c_vp, _c_vp]),
    "dss_blend_backward": (_c_int, [_c_vp] * 10 + [_c_int, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp]),
    "dss_blend_backward_scatter": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [_c_i64, _c_vp, _c_vp]),
}


def load():
    """Load libdss_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "dss_amd: %s not found. Build it with `make -C dss_amd/csrc` (or "
                    "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
                    % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


WS_ORDER_SAVE, WS_ORDER_REUSE = 0x10, 0x20   # include/dss_hip.h DSS_WS_ORDER_* (flags of workspace_state)
WS_BAND_OUTPUTS = 0x40   # DSS_WS_BAND_OUTPUTS: ellipse / scaler / cutoff only for the splats that meet the row band
OPT_LEAN_WORKSPACE, OPT_BACKWARD_TPW, OPT_BACKWARD_ADDR64, OPT_BACKWARD_FUSED, OPT_KNN_QUERY = 0, 1, 2, 3, 4   # include/dss_hip.h DSS_OPT_*


def set_option(option: int, value: int) -> int:
    """dss_set_option (explicit process-wide tuning hints; the library reads no environment variable). -> old value"""
    lib = load()
    old = lib.dss_get_option(option)
    check(lib.dss_set_option(option, value), "dss_set_option")
    return old


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dss_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("dss_amd: %s is on %s; the HIP path needs GPU tensors (no CPU fallback)"
                           % (name, t.device))
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("dss_amd: %s must be %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


_ws_cache = {}


_clean_cache = {}


def clean_workspace(device, tag, nbytes: int) -> torch.Tensor:
    """Zero-initialised buffer owned by ONE entry point and ONE problem size (`tag`) on the current stream:
    the DSS_WS_CLEAN contract of include/dss_hip.h (the library leaves it zero-filled after every successful
    call, so the per-call memset launch is skipped).  Call `drop_clean_workspace` if the call fails."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _clean_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if len(_clean_cache) >= 16:  # problem sizes come and go (tests): keep the cache small
            _clean_cache.pop(next(iter(_clean_cache)))
        buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        _clean_cache[key] = buf
    return buf


def drop_clean_workspace(device, tag) -> None:
    _clean_cache.pop((device.index, torch.cuda.current_stream(device).cuda_stream, tag), None)


def workspace(device, nbytes: int) -> torch.Tensor:
    """Per-device, per-stream scratch buffer (grown on demand, reused across calls)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
