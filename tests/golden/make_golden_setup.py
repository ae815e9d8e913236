"""Generates tests/golden/ref_setup_teapot.npz by running the REFERENCE's own Python per-point setup
(`SurfaceSplatting._get_per_point_info`, /root/reference/DSS/core/rasterizer.py:525-565 with
:404-496 and :293-342) in this container.

The reference module cannot be imported as is: it depends on pytorch3d, frnn, torch_batch_svd, trimesh,
skimage ... which are absent.  None of those is needed by the functions pinned here except
  - `pytorch3d.ops.knn_points`  (K=7 neighbours for the variance scale h)  -> brute-force torch stand-in
  - `pytorch3d.ops.eyes`        (batched identity)                          -> trivial stand-in
  - `pytorch3d.ops.padded_to_packed` (drop the padding rows)                -> trivial stand-in
so every missing module is replaced by an auto-stub whose attributes are inert classes, with those two
functions filled in.  Cameras / point clouds are the minimal objects of dss_amd (same accessor names and
the pytorch3d matrix conventions).  The reference source files are imported from where they lie; nothing
is copied.

    python tests/golden/make_golden_setup.py
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys

sys.dont_write_bytecode = True  # the reference checkout is read-only: importing it must not leave __pycache__ there
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

STUB_ROOTS = ("pytorch3d", "frnn", "torch_batch_svd", "trimesh", "skimage", "plyfile", "imageio", "easydict",
              "pymeshlab", "open3d", "prefix_sum", "tensorboardX")


class _Inert:
    def __init__(self, *a, **k):
        pass


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (_Inert,), {})
        setattr(self, name, obj)
        return obj


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _StubFinder())
import pytorch3d.ops as ops3d  # noqa: E402  (the stub)


import collections  # noqa: E402
_KNN = collections.namedtuple("KNN", "dists idx knn")


def _knn_points(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, **kw):
    d = torch.cdist(p1.double(), p2.double()).pow(2).float()
    if lengths2 is not None:
        for b in range(p2.shape[0]):
            d[b, :, int(lengths2[b]):] = float("inf")
    vals, idx = d.topk(K, dim=-1, largest=False)
    nn = None
    if return_nn:  # (N,P1,K,3)
        nn = torch.stack([p2[b][idx[b]] for b in range(p2.shape[0])], 0)
    return _KNN(vals, idx, nn)


ops3d.knn_points = _knn_points
import pytorch3d.ops.utils as _ops_utils  # noqa: E402


def _convert_pointclouds_to_tensor(pcl):
    if torch.is_tensor(pcl):
        return pcl, torch.full((pcl.shape[0],), pcl.shape[1], dtype=torch.int64, device=pcl.device)
    return pcl.points_padded(), pcl.num_points_per_cloud()


_ops_utils.convert_pointclouds_to_tensor = _convert_pointclouds_to_tensor
ops3d.convert_pointclouds_to_tensor = _convert_pointclouds_to_tensor
import torch_batch_svd as _tbs  # noqa: E402  (the stub)


def _batch_svd(x):
    """torch_batch_svd.svd returns (U, S, V) with x = U diag(S) V^T: torch.linalg.svd stand-in (fp32, like the
    reference's CUDA batched SVD)."""
    U, S, Vh = torch.linalg.svd(x, full_matrices=False)
    return U, S, Vh.transpose(-1, -2)


_tbs.svd = _batch_svd


def _padded_to_packed(inputs, first_idxs, num_inputs):
    ends = list(first_idxs[1:].tolist()) + [int(num_inputs)]
    return torch.cat([inputs[b, : int(e) - int(f)] for b, (f, e) in enumerate(zip(first_idxs.tolist(), ends))], dim=0)


ops3d.padded_to_packed = _padded_to_packed
ops3d.eyes = lambda dim, N, device=None, dtype=torch.float32: torch.eye(dim, device=device, dtype=dtype)[None].expand(N, dim, dim).clone()
import pytorch3d.renderer.points.rasterize_points as _rp  # noqa: E402
_rp.kMaxPointsPerBin = 22

sys.path.insert(0, "/root/reference")
import DSS  # noqa: E402  (the reference package; its compiled extension DSS._C is CUDA-only -> inert stub)
DSS._C = _StubModule("DSS._C")
sys.modules["DSS._C"] = DSS._C
ref_rast = importlib.import_module("DSS.core.rasterizer")  # the UNMODIFIED reference module

import scenes  # noqa: E402
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform  # noqa: E402
from dss_amd.cloud import PointClouds3D  # noqa: E402


def main():
    torch.manual_seed(0)
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    S = 256
    out = {}
    for tag, az in (("1cam", [45.0]), ("3cam", [10.0, 130.0, 250.0])):
        R, T = look_at_view_transform(2.0, 30.0, az)
        cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T)
        N = len(az)
        cloud = PointClouds3D([torch.from_numpy(pts)] * N, [torch.from_numpy(nrm)] * N)
        for mode, kw in (("global", dict(Vrk_invariant=True, Vrk_isotropic=False)),
                         ("iso", dict(Vrk_invariant=False, Vrk_isotropic=True)),
                         ("aniso", dict(Vrk_invariant=False, Vrk_isotropic=False))):
            st = ref_rast.PointsRasterizationSettings(cutoff_threshold=1.0, image_size=S, antialiasing_sigma=1.0, **kw)
            rast = ref_rast.SurfaceSplatting(cameras=cams, raster_settings=st, frnn_radius=-1)
            rast.cameras, rast._Vrk_h = cams, None  # (pytorch3d PointsRasterizer.__init__ would set .cameras)
            with torch.no_grad():
                info = rast._get_per_point_info(cloud, cameras=cams, raster_settings=st)
            for k, v in info.items():
                out["%s_%s_%s" % (tag, mode, k)] = v.numpy().astype(np.float32)
            if mode == "aniso" and tag == "1cam":  # the intermediate the new code is pinned on
                with torch.no_grad():
                    Vr, Sk = rast._compute_anisotropic_Vrk(cloud)
                out["aniso_Vr"], out["aniso_Sk"] = Vr.numpy().astype(np.float32), Sk.numpy().astype(np.float32)
        out[tag + "_M"] = cams.get_full_projection_transform().get_matrix().numpy()
        out[tag + "_V"] = cams.get_world_to_view_transform().get_matrix().numpy()
    out["points"], out["normals"], out["S"] = pts, nrm, np.int32(S)
    np.savez_compressed(os.path.join(HERE, "ref_setup_teapot.npz"), **out)
    for k, v in out.items():
        print(k, getattr(v, "shape", v))


def lighting_vectors():
    """tests/golden/ref_lighting.npz: the reference's own `diffuse` / `specular` (DSS/core/lighting.py:10-172) on
    packed inputs, combined like LightingTexture.forward (texture.py:118-122), for two clouds with two lights each,
    as point lights (direction = location - point, lighting.py:270-276) and as directional lights.
    `convert_to_tensors_and_broadcast` (pytorch3d, absent) is replaced by a stand-in that broadcasts dim 0."""
    import pytorch3d.renderer as p3r  # the stub

    def _ctb(*args, device="cpu", dtype=torch.float32):
        ts = [a if torch.is_tensor(a) else torch.as_tensor(a, dtype=dtype) for a in args]
        n = max(t.shape[0] if t.dim() > 0 else 1 for t in ts)
        out = []
        for t in ts:
            if t.dim() == 0:
                t = t.reshape(1)
            out.append(t.expand((n,) + tuple(t.shape[1:])) if t.shape[0] == 1 and n > 1 else t)
        return out

    p3r.convert_to_tensors_and_broadcast = _ctb
    import pytorch3d.renderer.lighting as _pl  # stub submodule: its DirectionalLights / PointLights are inert bases
    p3r.lighting = _pl
    lighting = importlib.import_module("DSS.core.lighting")
    lighting.convert_to_tensors_and_broadcast = _ctb
    g = torch.Generator().manual_seed(7)
    num = [700, 500]
    P, N, L = sum(num), 2, 2
    pts = torch.randn(P, 3, generator=g) * 0.6
    nrm = torch.randn(P, 3, generator=g)
    nrm[::7] *= 30.0                      # un-normalised normals, like bunny-8000.ply
    rgb = torch.rand(P, 3, generator=g)
    batch = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(num)])
    amb = torch.rand(N, 1, 3, generator=g) * 0.5
    kd = torch.rand(N, L, 3, generator=g)
    ks = torch.rand(N, L, 3, generator=g)
    vec = torch.randn(N, L, 3, generator=g) * 2.0
    cam = torch.randn(N, 3, generator=g) * 3.0
    out = dict(points=pts.numpy(), normals=nrm.numpy(), rgb=rgb.numpy(), num=np.asarray(num, np.int64),
               ambient=amb.numpy(), diffuse_color=kd.numpy(), specular_color=ks.numpy(), light_vec=vec.numpy(),
               cam_center=cam.numpy(), shininess=np.float32(24.0))
    for tag in ("point", "directional"):
        direction = vec[batch] - pts[:, None, :] if tag == "point" else vec[batch]
        dif = lighting.diffuse(normals=nrm, color=kd[batch], direction=direction)
        spc = lighting.specular(points=pts, normals=nrm, direction=direction, color=ks[batch],
                                camera_position=cam[batch], shininess=torch.full((P,), 24.0))
        ambient = amb.sum(1)[batch]
        out[tag + "_diffuse"], out[tag + "_specular"] = dif.numpy(), spc.numpy()
        out[tag + "_shaded"] = (rgb * (ambient + dif) + spc).numpy()
    np.savez_compressed(os.path.join(HERE, "ref_lighting.npz"), **out)
    print("ref_lighting.npz", {k: getattr(v, "shape", v) for k, v in out.items()})


class _Captured(Exception):
    pass


def backward_radius_vectors():
    """Runs the Python half of the reference's EllipticalRasterizer.backward (rasterizer.py:853-913: visible
    set from the fragments, per-cloud median of the visible radii times radii_s, 2-D grid parameters) up to
    its first third-party CUDA call (frnn._C.insert_points_cuda), which is replaced by a stub that captures
    the arguments: grid_params[:, 2] = 1/cell with cell = search_radius/2 (RADIUS_CELL_RATIO = 2,
    rasterizer.py:877, 897-901), so the search radius and the visible counts are recovered exactly."""
    import frnn  # the stub
    captured = {}

    def insert_points_cuda(pts2d, lengths, grid_params, *rest):
        captured["lengths"] = lengths.clone()
        captured["grid_params"] = grid_params.clone()
        raise _Captured()

    frnn._C.insert_points_cuda = insert_points_cuda

    def packed_to_padded(inputs, first_idxs, max_size):
        n = first_idxs.shape[0]
        ends = list(first_idxs[1:].tolist()) + [inputs.shape[0]]
        out = inputs.new_zeros((n, int(max_size)) + tuple(inputs.shape[1:]))
        for b, (f, e) in enumerate(zip(first_idxs.tolist(), ends)):
            out[b, : e - f] = inputs[f:e]
        return out

    ops3d.packed_to_padded = packed_to_padded
    import prefix_sum  # the stub (imported inside backward, rasterizer.py:872)
    prefix_sum.prefix_sum_cuda = lambda *a, **k: None
    out = {}
    for name in ("ref_random64x2", "ref_teapot256", "ref_ties32"):
        z = np.load(os.path.join(HERE, name + ".npz"))
        t = lambda k: torch.from_numpy(z[k])
        for radii_s in (5.0, 1.0):
            ctx = types.SimpleNamespace(radii_backward_scaler=radii_s, depth_merging_threshold=float(z["thr"]),
                                        saved_tensors=(t("points"), t("ellipse"), t("cutoff"), t("radii"), t("ref_idx"),
                                                       t("ref_zbuf")[..., 0].clone(), t("first_idx"), t("num_pts")))
            try:
                ref_rast.EllipticalRasterizer.backward(ctx, None, torch.zeros_like(t("ref_zbuf")), None,
                                                       torch.from_numpy(z["grad_occ"]))
                raise RuntimeError("stub was not reached")
            except _Captured:
                pass
            gp = captured["grid_params"].numpy()
            out["%s_s%g_num_visible" % (name, radii_s)] = captured["lengths"].numpy().astype(np.int64)
            out["%s_s%g_inv_cell" % (name, radii_s)] = gp[:, 2].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_backward_radius.npz"), **out)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    lighting_vectors()
    main()
    backward_radius_vectors()
