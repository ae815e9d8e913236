"""Generates tests/golden/ref_fast_backward.npz: the reference's fast occupancy backward, EXECUTED.

What runs (in this container, CPU only):
  * the UNMODIFIED Python of `EllipticalRasterizer.backward` (/root/reference/DSS/core/rasterizer.py:787-977,
    `backward_occ_fast = True`), imported from where it lies;
  * inside it, `_C._splat_points_occ_fast_cuda_backward` = the UNMODIFIED CUDA kernel
    `RasterizePointsBackwardCudaFastKernel` (DSS/csrc/rasterize_points_backward.cu:30-212) compiled for the host by
    oracle/ref_cuda_host.cpp (`make -C oracle ref` -> oracle/_ref/libref_cuda_host.so), launched 1024 x 64 like
    rasterize_points_backward.cu:306-307;
  * `_C._backward_zbuf` = the reference CPU build (oracle/_ref/dss_ref_cpu);
  * separately (`*_slowcuda_grad`): the older `_C._splat_points_occ_backward` in its CUDA form,
    RasterizePointsOccBackwardCudaKernel (rasterize_points.cu:672-757), host-compiled the same way, on all points with
    the radii_s stored in the scene's golden file (ragged3: 2.0).
Third-party pieces that are absent here are replaced by numpy stand-ins of their published behaviour:
  * frnn._C.insert_points_cuda / counting_sort_cuda (lxxue/FRNN, 2-D grid): cell = floor((p - min) * delta) per axis
    clamped to [0, res-1], linear id x*res_y + y (the id the consumer kernel computes, rasterize_points_backward.cu:119),
    slot = arrival order inside the cell; sorted[off[cell] + slot] = point.  Only the SET of points per cell matters.
  * prefix_sum.prefix_sum_cuda (lxxue/prefix_sum): exclusive scan of the cell counts.
  * pytorch3d.ops.packed_to_padded / padded_to_packed: (un)padding of packed rows.

Known reference behaviour recorded with the vectors (and asserted by tests/test_oracle_pinning.py):
  * `lastcell`: the kernel ends the LAST grid cell of cloud n at `num_points_per_cloud[n]` instead of
    `first_idx[n] + num_points_per_cloud[n]` (rasterize_points_backward.cu:124-126), so for clouds n >= 1 the points of
    that cell receive no gradient.  The oracle / HIP path give them their gradient.
  * `center_*`: a point exactly on a pixel centre (d^2 == 0): device eps_denom returns 0 (sign(0) = 0,
    rasterization_utils.cuh:38-43) -> 0/0 = NaN in the reference; the oracle / HIP path contribute 0.

    python tests/golden/make_golden_fast_backward.py
"""
import ctypes
import os
import sys

sys.dont_write_bytecode = True  # the reference checkout is read-only: importing it must not leave __pycache__ there
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden_setup as mgs  # noqa: E402  (installs the stub importer, imports the reference rasterizer module)

import oracle  # noqa: E402
import scenes  # noqa: E402

ref_rast, ops3d = mgs.ref_rast, mgs.ops3d
HOST = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_cuda_host.so"))
REF = oracle.ref()
assert REF is not None, "build oracle/_ref first: make -C oracle ref"
_p = lambda t: ctypes.c_void_p(t.data_ptr())
captured = {}


# ---- third-party stand-ins (numpy / torch, CPU) ---------------------------------------------------------------
def packed_to_padded(inputs, first_idxs, max_size):
    n = first_idxs.shape[0]
    ends = list(first_idxs[1:].tolist()) + [inputs.shape[0]]
    out = inputs.new_zeros((n, int(max_size)) + tuple(inputs.shape[1:]))
    for b, (f, e) in enumerate(zip(first_idxs.tolist(), ends)):
        out[b, : e - f] = inputs[f:e]
    return out


def insert_points_cuda(pts2d, lengths, grid_params, cnt, cell, slot, G):
    gp = grid_params.numpy()
    for n in range(pts2d.shape[0]):
        L = int(lengths[n])
        p = pts2d[n, :L].numpy()
        mn, delta = gp[n, 0:2].astype(np.float32), np.float32(gp[n, 2])
        res = gp[n, 3:5].astype(np.int64)
        g = np.floor((p - mn[None, :]) * delta).astype(np.int64)
        g = np.clip(g, 0, res[None, :] - 1)
        c = g[:, 0] * res[1] + g[:, 1]
        assert c.max() < int(gp[n, 5])
        counts = np.zeros(G, np.int64)
        s = np.empty(L, np.int64)
        for i, ci in enumerate(c):       # arrival order = point order (any order is a valid atomicAdd outcome)
            s[i] = counts[ci]
            counts[ci] += 1
        cnt[n] = torch.from_numpy(counts.astype(np.int32))
        cell[n, :L] = torch.from_numpy(c.astype(np.int32))
        slot[n, :L] = torch.from_numpy(s.astype(np.int32))
    captured["cell"], captured["lengths"], captured["grid_params"] = cell.clone(), lengths.clone(), grid_params.clone()


def prefix_sum_cuda(counts, total, out):
    total = int(total)
    c = counts[:total].to(torch.int64)
    out[:total] = (torch.cumsum(c, 0) - c).to(out.dtype)


def counting_sort_cuda(pts2d, lengths, cell, slot, off, sorted_pts, sorted_idx):
    for n in range(pts2d.shape[0]):
        L = int(lengths[n])
        dst = (off[n][cell[n, :L].long()] + slot[n, :L]).long()
        sorted_pts[n, dst] = pts2d[n, :L]
        sorted_idx[n, dst] = torch.arange(L, dtype=sorted_idx.dtype)


def fast_cuda_backward(points_sorted, radii_sorted, rs, grad_occ, num_points, first_idx, grid_off, grid_params):
    """`DSS._C._splat_points_occ_fast_cuda_backward` (ext.cpp:14): the host-compiled reference kernel."""
    points_sorted, radii_sorted = points_sorted.contiguous().float(), radii_sorted.contiguous().float()
    rs, grad_occ = rs.contiguous().float(), grad_occ.contiguous().float()
    num_points, first_idx = num_points.contiguous().long(), first_idx.contiguous().long()
    grid_off, grid_params = grid_off.contiguous().int(), grid_params.contiguous().float()
    N, H, W = grad_occ.shape
    P, G = points_sorted.shape[0], grid_off.shape[1]
    # the reference's last-cell end (rasterize_points_backward.cu:124-126) must stay an EMPTY range for clouds n >= 1
    for n in range(1, N):
        total = int(grid_params[n, 5])
        assert int(num_points[n]) <= int(grid_off[n, total - 1]), "last-cell bug would walk foreign points"
    grad = torch.empty((P, 2), dtype=torch.float32)
    HOST.ref_fast_backward(_p(points_sorted), _p(radii_sorted), _p(rs), _p(num_points), _p(first_idx), _p(grid_off),
                           _p(grid_params), _p(grad_occ), N, H, W, G, ctypes.c_int64(P), _p(grad))
    return grad


import frnn  # noqa: E402  (the stub)
import prefix_sum  # noqa: E402  (the stub)
frnn._C.insert_points_cuda = insert_points_cuda
frnn._C.counting_sort_cuda = counting_sort_cuda
prefix_sum.prefix_sum_cuda = prefix_sum_cuda
ops3d.packed_to_padded = packed_to_padded
ref_rast.frnn = frnn
ref_rast._C._splat_points_occ_fast_cuda_backward = fast_cuda_backward
ref_rast._C._backward_zbuf = REF._backward_zbuf


def reference_backward(sc, idx, zbuf, grad_occ, grad_zbuf, radii_s, thr):
    """-> (grad (P,3) of the reference, lastcell mask (P,) bool)."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    ctx = types.SimpleNamespace(radii_backward_scaler=radii_s, depth_merging_threshold=float(thr),
                                saved_tensors=(t(sc["points"]), t(sc["ellipse"]), t(sc["cutoff"]), t(sc["radii"]), t(idx),
                                               t(zbuf)[..., 0].clone(), t(sc["first_idx"]), t(sc["num_pts"])))
    out = ref_rast.EllipticalRasterizer.backward(ctx, None, t(grad_zbuf), None, t(grad_occ))
    grad = out[0].numpy()
    # points of the last grid cell of clouds n >= 1 (visible order -> packed order)
    P = sc["points"].shape[0]
    vis = oracle.visibility(idx, P)
    lastcell = np.zeros(P, bool)
    vis_ids = np.nonzero(vis)[0]
    lengths, off = captured["lengths"].numpy(), 0
    for n in range(len(lengths)):
        L = int(lengths[n])
        if n >= 1:
            total = int(captured["grid_params"][n, 5])
            lastcell[vis_ids[off:off + L][captured["cell"][n, :L].numpy() == total - 1]] = True
        off += L
    return grad.astype(np.float32), lastcell


def slow_cuda_backward(sc, grad_occ, radii_s):
    """`DSS._C._splat_points_occ_backward` on CUDA tensors = RasterizePointsOccBackwardCudaKernel
    (rasterize_points.cu:672-757), host-compiled and executed: every point handed in takes part."""
    pts, radii = (torch.from_numpy(np.ascontiguousarray(sc[k], np.float32)) for k in ("points", "radii"))
    first, num = (torch.from_numpy(np.ascontiguousarray(sc[k], np.int64)) for k in ("first_idx", "num_pts"))
    gocc = torch.from_numpy(np.ascontiguousarray(grad_occ, np.float32))
    N, H, W = gocc.shape
    grad = torch.empty((pts.shape[0], 2), dtype=torch.float32)
    HOST.ref_slow_backward_cuda(_p(pts), _p(radii), _p(first), _p(num), ctypes.c_float(radii_s), N, H, W, _p(gocc),
                                ctypes.c_int64(pts.shape[0]), _p(grad))
    return grad.numpy()


def main():
    out = {}
    # scenes with committed forward goldens (ref_idx = output of the compiled reference CPU rasterizer)
    for name in ("ref_random64x2", "ref_teapot256", "ref_ties32"):
        z = np.load(os.path.join(HERE, name + ".npz"))
        sc = {k: z[k] for k in ("points", "ellipse", "cutoff", "radii", "first_idx", "num_pts")}
        for radii_s in (5.0, 1.0):
            grad, lastcell = reference_backward(sc, z["ref_idx"], z["ref_zbuf"], z["grad_occ"], z["grad_zbuf"], radii_s,
                                                z["thr"])
            out["%s_s%g_grad" % (name, radii_s)] = grad
            out["%s_s%g_lastcell" % (name, radii_s)] = lastcell
        out[name + "_slowcuda_grad"] = slow_cuda_backward(sc, z["grad_occ"], float(z["radii_s"]))
    # three ragged clouds in one batch (exercises first_idx handling and the last-cell behaviour twice)
    S, K, thr = 80, 4, 0.3
    sc = scenes.random_splats(800, S, 3, seed=21)
    P = sc["points"].shape[0]
    sc["first_idx"] = np.array([0, 700, 1900], np.int64)
    sc["num_pts"] = np.array([700, 1200, P - 1900], np.int64)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sc[k]))
    idx, zbuf, qv, occ = REF.splat_points(t("points"), t("ellipse"), t("cutoff"), t("radii"), t("first_idx"), t("num_pts"),
                                          thr, S, K, 0, 0)
    rng = np.random.default_rng(4)
    gocc = rng.standard_normal((3, S, S)).astype(np.float32)
    gocc[rng.random((3, S, S)) < 0.3] = 0.0
    gz = rng.standard_normal((3, S, S, K)).astype(np.float32)
    for k in ("points", "ellipse", "cutoff", "radii", "first_idx", "num_pts"):
        out["ragged3_" + k] = sc[k]
    out["ragged3_S"], out["ragged3_K"], out["ragged3_thr"] = np.int32(S), np.int32(K), np.float32(thr)
    out["ragged3_idx"], out["ragged3_zbuf"] = idx.numpy(), zbuf.numpy()
    out["ragged3_grad_occ"], out["ragged3_grad_zbuf"] = gocc, gz
    for radii_s in (5.0, 2.0):
        grad, lastcell = reference_backward(sc, idx.numpy(), zbuf.numpy(), gocc, gz, radii_s, thr)
        out["ragged3_s%g_grad" % radii_s], out["ragged3_s%g_lastcell" % radii_s] = grad, lastcell
    out["ragged3_slowcuda_grad"] = slow_cuda_backward(sc, gocc, 2.0)
    # a point exactly on a pixel centre: d^2 == 0 -> NaN in the reference (known divergence)
    S = 16
    c = np.float32(-1) + np.float32(2 * 5 + 1) / np.float32(S)        # centre of NDC index 5 (exact in fp32)
    one = dict(points=np.array([[c, c, 1.0], [0.3, -0.2, 1.5]], np.float32),
               ellipse=np.array([[30.0, 0.0, 30.0]] * 2, np.float32), cutoff=np.ones(2, np.float32),
               radii=np.full((2, 2), 0.19, np.float32), first_idx=np.zeros(1, np.int64), num_pts=np.array([2], np.int64))
    t = lambda k: torch.from_numpy(one[k])
    idx, zbuf, qv, occ = REF.splat_points(t("points"), t("ellipse"), t("cutoff"), t("radii"), t("first_idx"), t("num_pts"),
                                          0.05, S, 2, 0, 0)
    gocc = np.ones((1, S, S), np.float32)
    grad, _ = reference_backward(one, idx.numpy(), zbuf.numpy(), gocc, np.zeros((1, S, S, 2), np.float32), 3.0, 0.05)
    for k, v in one.items():
        out["center_" + k] = v
    out["center_idx"], out["center_grad_occ"], out["center_grad"] = idx.numpy(), gocc, grad
    np.savez_compressed(os.path.join(HERE, "ref_fast_backward.npz"), **out)
    for k, v in out.items():
        if k.endswith("_grad"):
            print(k, v.shape, "finite" if np.isfinite(v).all() else "HAS NaN/inf", float(np.nanmax(np.abs(v))))
        if k.endswith("_lastcell"):
            print(k, int(v.sum()), "points in the last cell of clouds n>=1")


if __name__ == "__main__":
    main()
