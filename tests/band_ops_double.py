"""TEST DOUBLE (CPU): the `dss_amd.ops` entry points `dss_amd.sharded.RowShardedRender` calls, with ROW BANDS, answered by the
oracle -- so that the engine's exchange logic (which rank renders what, what travels in which collective, how the partial
results are put together) runs over gloo on the GPU-less build container.  The band-aware counterparts of
tests/ref_loop/oracle_ops.py; never imported by the product."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_loop"))
import oracle  # noqa: E402
import oracle_ops  # noqa: E402


def owned_rows(rows, S):
    if rows is None:
        return list(range(S))
    c = rows[2] if len(rows) > 2 else 1
    if c > 1:
        return [r + i for r in range(rows[0], rows[1], 8 * c) for i in range(8) if r + i < rows[1]]
    return list(range(rows[0], rows[1]))


def render_forward(world, normals, h, M, V, znear, zfar, first, num, features, image_size, points_per_pixel, cutoff_threshold,
                   depth_merging_thres, antialiasing_sigma=1.0, backface_culling=False, shared_cloud=False, rows=None,
                   out_image=None, out_visible=None, vr6=None, frame_normals=None, want_zbuf=True, workspace_state=1,
                   order_refresh=0, band_outputs_only=False, point_outputs=None):
    o = oracle_ops.render_forward(world, normals, h, M, V, znear, zfar, first, num, features, image_size, points_per_pixel,
                                  cutoff_threshold, depth_merging_thres, antialiasing_sigma, backface_culling, shared_cloud)
    S, P = int(image_size), o["pts_screen"].shape[0]
    ri = owned_rows(rows, S)
    for k in ("idx", "zbuf", "qvalue", "occupancy", "image", "wsum"):
        o[k] = o[k][:, ri].contiguous()
    # (utils/__init__.py:320-340 on the band's rows; oracle.visibility takes square images only)
    ids = o["idx"].numpy()
    v8 = np.zeros((P,), np.uint8)
    v8[np.unique(ids[ids >= 0])] = 1
    vis = torch.from_numpy(v8)
    if out_visible is not None:
        out_visible.copy_(vis)
        vis = out_visible
    if out_image is not None:
        out_image.copy_(o["image"])
        o["image"] = out_image
    o["visible"] = vis.view(torch.bool) if vis.dtype == torch.uint8 else vis
    return o


def _centre_row(py, S):   # raster_backward.hip centre_image_row
    fy = (1.0 - py.astype(np.float32)) * np.float32(0.5) * np.float32(S)
    return np.clip(fy.astype(np.int64), 0, S - 1)


def render_backward(grad_out, idx, qvalue, wsum, scaler, points, radii, visible, first, num, radii_s, clip=-1.0,
                    with_features=True, return_rs=False, image_size=None, rows=None, out=None, gather_only_rs=None,
                    project=None, grad_out_full=None, grad_occ_full=None):
    N, H, W, K = idx.shape
    S, P = int(image_size) if image_size is not None else W, points.shape[0]
    ri = owned_rows(rows, S)
    # (the oracle's blend takes square images: the band embedded in an otherwise empty full image gives the band's sums)
    full_idx = torch.full((N, S, S, K), -1, dtype=idx.dtype)
    full_q = torch.full((N, S, S, K), -1.0)
    full_g = torch.zeros((N, S, S, grad_out.shape[-1]))
    full_idx[:, ri], full_q[:, ri], full_g[:, ri] = idx, qvalue, grad_out
    gf, gocc_full = oracle_ops.blend_backward(full_g, full_idx, full_q, scaler, P)
    gocc_band = gocc_full[:, ri]
    vis = np.ascontiguousarray(visible.numpy().astype(bool))
    rs = oracle.backward_radius(oracle_ops._np(radii), vis, oracle_ops._np(first), oracle_ops._np(num), float(radii_s))
    pts = oracle_ops._np(points)
    owner = (grad_out_full is not None or grad_occ_full is not None) and len(ri) < S
    if owner:
        plane = oracle_ops._np(grad_out_full[..., -1] if grad_out_full is not None else grad_occ_full)
        mine = np.isin(_centre_row(pts[:, 1], S), np.asarray(ri))
        flags = vis & mine
    else:
        plane = np.zeros((N, S, S), np.float32)
        plane[:, ri] = oracle_ops._np(gocc_band)
        flags = vis
    g = oracle.occ_backward_fast(pts, oracle_ops._np(radii), flags, rs, np.ascontiguousarray(plane, np.float32),
                                 oracle_ops._np(first), oracle_ops._np(num))
    g = np.ascontiguousarray(g, np.float32)
    if g.shape[1] == 2:   # (the occupancy surrogate has no z component; the zbuf gradient is not on this path)
        g = np.concatenate([g, np.zeros((g.shape[0], 1), np.float32)], 1)
    g = torch.from_numpy(g)
    if clip is not None and clip > 0:
        nrm = g.norm(dim=1, keepdim=True)
        g = torch.where(nrm > clip, g * (clip / nrm.clamp(min=1e-30)), g)
    if out is not None:
        out[0].copy_(gf)
        out[1].copy_(g)
        gf, g = out
    return (gf, g, torch.from_numpy(rs)) if return_rs else (gf, g)


def project_backward(world, M, V, first, num, grad_screen, valid, shared_cloud=False, clip=-1.0, grad_features=None, out=None):
    gw = oracle_ops.project_backward(world, M, V, first, num, grad_screen, valid, shared_cloud, clip)
    if out is not None:
        out[0].copy_(gw)
        gw = out[0]
    if grad_features is None:
        return gw
    N, Pw = first.shape[0], world.shape[0]
    gfw = grad_features.view(N, Pw, -1).sum(0) if shared_cloud else grad_features.clone()
    if out is not None and out[1] is not None:
        out[1].copy_(gfw)
        gfw = out[1]
    return gw, gfw


def gather_rows(src, row_pos, n_images, rows, row_floats, out=None):
    res = src.reshape(-1, n_images, row_floats).index_select(0, row_pos.long()).permute(1, 0, 2).contiguous()
    if out is not None:
        out.view(n_images, rows, row_floats).copy_(res)
        return out
    return res


def install(ops_module) -> None:
    oracle_ops.install(ops_module)
    for name in ("render_forward", "render_backward", "project_backward", "gather_rows"):
        setattr(ops_module, name, globals()[name])
