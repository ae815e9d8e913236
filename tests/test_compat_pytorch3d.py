"""The `compat/pytorch3d` namespace package: the names yifita/DSS imports from pytorch3d, checked against
closed forms / scipy / dss_amd's own host mirrors on the CPU, and -- on the GPU -- that objects of the shim drive the
drop-in renderer built the way `config.create_renderer` (config.py:241-261) builds it."""
import importlib
import io
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "compat") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "compat"))

import pytorch3d  # noqa: E402  (the stand-in)
from pytorch3d.io import load_obj, load_ply, save_obj, save_ply  # noqa: E402
from pytorch3d.loss import chamfer_distance  # noqa: E402
from pytorch3d.ops import (convert_pointclouds_to_tensor, is_pointclouds, knn_gather, knn_points, packed_to_padded,  # noqa: E402
                           padded_to_packed, sample_points_from_meshes)
from pytorch3d.renderer import (FoVOrthographicCameras, FoVPerspectiveCameras, PerspectiveCameras, TensorProperties,  # noqa: E402
                                convert_to_tensors_and_broadcast, look_at_rotation, look_at_view_transform)
from pytorch3d.renderer.compositing import norm_weighted_sum, weighted_sum  # noqa: E402
from pytorch3d.structures import Pointclouds, list_to_padded, padded_to_list  # noqa: E402
from pytorch3d.transforms import Rotate, RotateAxisAngle, Scale, Transform3d, Translate  # noqa: E402
from pytorch3d.utils import ico_sphere  # noqa: E402


def test_it_is_the_stand_in_and_not_a_real_install():
    assert "dss_amd.compat" in pytorch3d.__version__ and pytorch3d.__file__.startswith(os.path.join(ROOT, "compat"))


def test_transform3d_row_vector_composition_and_inverses():
    t = Transform3d().scale(2.0).translate(1.0, 0.0, 0.0)       # applied left to right: scale, then translate
    p = torch.tensor([[1.0, 1.0, 1.0]])
    assert torch.allclose(t.transform_points(p), torch.tensor([[3.0, 2.0, 2.0]]))
    assert torch.allclose(t.inverse().transform_points(t.transform_points(p)), p, atol=1e-6)
    assert torch.allclose(t.get_matrix()[0, 3, :3], torch.tensor([1.0, 0.0, 0.0]))  # translation = last ROW
    r = RotateAxisAngle(90.0, "Z")
    assert torch.allclose(r.transform_points(torch.tensor([[1.0, 0.0, 0.0]])), torch.tensor([[0.0, 1.0, 0.0]]), atol=1e-6)
    assert torch.allclose(r.inverse().get_matrix(), r.get_matrix().transpose(1, 2), atol=1e-7)
    # normals transform with the inverse transpose: a non-uniform scale keeps them perpendicular to the surface
    s = Scale(1.0, 2.0, 4.0)
    tangent, normal = torch.tensor([[1.0, 1.0, 0.0]]), torch.tensor([[1.0, -1.0, 0.0]])
    tt = s.transform_points(tangent) - s.transform_points(torch.zeros(1, 3))
    assert abs(float((tt * s.transform_normals(normal)).sum())) < 1e-6
    batch = Translate(torch.tensor([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]))
    assert len(batch) == 2 and len(batch.compose(Rotate(torch.eye(3)))) == 2
    assert torch.allclose(batch.stack(batch).get_matrix()[2:], batch.get_matrix())


def test_cameras_follow_the_documented_conventions_and_agree_with_the_host_mirror():
    from dss_amd import cameras as host
    R, T = look_at_view_transform(dist=2.7, elev=20.0, azim=30.0)
    R2, T2 = host.look_at_view_transform(2.7, 20.0, 30.0)
    assert torch.allclose(R, R2, atol=1e-6) and torch.allclose(T, T2, atol=1e-6)
    cam = FoVPerspectiveCameras(R=R, T=T, znear=0.1, zfar=100.0, fov=60.0)
    cam2 = host.FoVPerspectiveCameras(R=R2, T=T2, znear=0.1, zfar=100.0, fov=60.0)
    assert torch.allclose(cam.get_full_projection_transform().get_matrix(), cam2.get_full_projection_transform().get_matrix(),
                          atol=1e-6)
    # the look-at target projects to the image centre; the camera centre is at distance `dist` from it
    ndc = cam.transform_points(torch.zeros(1, 1, 3))
    assert torch.allclose(ndc[0, 0, :2], torch.zeros(2), atol=1e-6)
    assert abs(float(cam.get_camera_center().norm()) - 2.7) < 1e-5
    # a camera on +z looking at the origin sees world +x on its right: NDC x is NEGATIVE there (pytorch3d NDC: +x left, +y up)
    cam0 = FoVPerspectiveCameras(*[], R=look_at_view_transform(2.0, 0.0, 0.0)[0], T=look_at_view_transform(2.0, 0.0, 0.0)[1])
    x = cam0.transform_points(torch.tensor([[[0.3, 0.0, 0.0], [0.0, 0.3, 0.0]]]))
    assert x[0, 0, 0] < 0 and x[0, 1, 1] > 0
    # depth range: znear -> 0, zfar -> 1
    view = cam0.get_world_to_view_transform().transform_points(torch.tensor([[[0.0, 0.0, 1.0]]]))   # 1 unit in front
    assert abs(float(view[0, 0, 2]) - 1.0) < 1e-6
    z = cam0.get_projection_transform().transform_points(torch.tensor([[[0.0, 0.0, 1.0], [0.0, 0.0, 100.0]]]))[0, :, 2]
    assert abs(float(z[0])) < 1e-6 and abs(float(z[1]) - 1.0) < 1e-5
    # unproject inverts project
    pts = torch.rand(1, 5, 3) - 0.5
    proj = cam.transform_points(pts)
    assert torch.allclose(cam.unproject_points(proj, world_coordinates=True, scaled_depth_input=True), pts, atol=1e-4)
    ortho = FoVOrthographicCameras(R=R, T=T)
    assert torch.allclose(ortho.unproject_points(ortho.transform_points(pts), scaled_depth_input=True), pts, atol=1e-5)
    pin = PerspectiveCameras(focal_length=2.0, R=R, T=T)
    v = pin.get_world_to_view_transform().transform_points(pts)
    assert torch.allclose(pin.transform_points(pts)[..., :2], 2.0 * v[..., :2] / v[..., 2:], atol=1e-5)
    # rotation matrices are orthonormal, also when the view direction is almost parallel to `up`
    Rd = look_at_rotation(((0.05, 3.0, 0.0),), up=((0.0, 1.0, 0.0),))
    assert torch.allclose(Rd[0] @ Rd[0].t(), torch.eye(3), atol=1e-5)
    # R, T kwargs override and are remembered (trainer.py:264 assigns cameras.R / cameras.T directly)
    cam.R, cam.T = R2, T2
    cam._N = 1
    assert cam.to("cpu") is cam and len(cam.clone()) == 1


def test_tensor_properties_broadcast_and_gather():
    tp = TensorProperties(a=((1.0, 2.0, 3.0),), b=torch.ones(4, 2), c=None, name="x", k=2.0)
    assert len(tp) == 4 and tp.a.shape == (4, 3) and tp.b.shape == (4, 2) and tp.c is None and tp.name == "x" and tp.k.shape == (4,)
    g = TensorProperties(a=torch.arange(3.0)[:, None]).gather_props(torch.tensor([2, 2, 0, 1]))
    assert g.a[:, 0].tolist() == [2.0, 2.0, 0.0, 1.0]
    with pytest.raises(ValueError):
        convert_to_tensors_and_broadcast(torch.ones(2, 3), torch.ones(3, 3))


def test_pointclouds_list_padded_packed_views_and_private_fields():
    a, b = torch.rand(5, 3), torch.rand(2, 3)
    pc = Pointclouds([a, b], normals=[a + 1, b + 1], features=[torch.rand(5, 4), torch.rand(2, 4)])
    assert len(pc) == 2 and not pc.isempty() and pc._P == 5 and pc._C == 4 and not pc.equisized
    assert pc.points_padded().shape == (2, 5, 3) and torch.equal(pc.points_padded()[1, 2:], torch.zeros(3, 3))
    assert torch.equal(pc.points_packed(), torch.cat([a, b])) and pc.cloud_to_packed_first_idx().tolist() == [0, 5]
    assert pc.packed_to_cloud_idx().tolist() == [0] * 5 + [1] * 2 and pc.num_points_per_cloud().tolist() == [5, 2]
    assert torch.equal(pc.points_padded().reshape(-1, 3)[pc.padded_to_packed_idx()], pc.points_packed())
    assert torch.equal(pc.normals_packed(), torch.cat([a, b]) + 1) and pc.features_padded().shape == (2, 5, 4)
    assert pc._normals_packed is not None and pc._points_list is not None and pc._features_padded is not None
    # from a padded tensor; gradients reach the source through every view
    src = torch.rand(2, 6, 3, requires_grad=True)
    pp = Pointclouds(src, normals=torch.rand(2, 6, 3))
    assert pp.equisized and pp.points_packed().shape == (12, 3)
    pp.points_packed().sum().backward()
    assert torch.equal(src.grad, torch.ones_like(src))
    # in-place offset, extend, indexing, bounding boxes, update_padded
    before = pc.points_packed().clone()
    pc.offset_(torch.ones_like(before))
    assert torch.allclose(pc.points_packed(), before + 1) and torch.allclose(pc.points_list()[1], b + 1)
    assert torch.allclose(pc.points_padded()[1, :2], b + 1)
    e = pc.extend(3)
    assert len(e) == 6 and e.num_points_per_cloud().tolist() == [5, 5, 5, 2, 2, 2]
    assert pc[1].points_packed().shape == (2, 3) and len(pc[[0, 1]]) == 2 and len(pc[torch.tensor([True, False])]) == 1
    assert torch.allclose(pc.get_bounding_boxes()[0, :, 0], (a + 1).min(0)[0])
    up = pc.update_padded(pc.points_padded() * 2)
    assert torch.allclose(up.points_packed(), pc.points_packed() * 2) and torch.equal(up.normals_packed(), pc.normals_packed())
    assert is_pointclouds(pc) and convert_pointclouds_to_tensor(pc)[1].tolist() == [5, 2]
    lst = padded_to_list(list_to_padded([a, b]), [5, 2])
    assert torch.equal(lst[0], a) and torch.equal(lst[1], b)
    empty = Pointclouds([torch.zeros(0, 3)])
    assert empty.isempty() and empty.points_packed().shape == (0, 3)
    assert Pointclouds([]).isempty()


def test_knn_matches_the_kd_tree_and_is_differentiable():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(0)
    a, b = rng.normal(size=(300, 3)).astype(np.float32), rng.normal(size=(40, 3)).astype(np.float32)
    p1 = torch.zeros(2, 300, 3)
    p1[0], p1[1, :40] = torch.from_numpy(a), torch.from_numpy(b)
    lengths = torch.tensor([300, 40])
    out = knn_points(p1, p1, lengths, lengths, K=6, return_nn=True)
    for n, pts in enumerate((a, b)):
        d, i = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=6)
        assert np.array_equal(out.idx[n, : len(pts)].numpy(), i)
        assert np.allclose(out.dists[n, : len(pts)].numpy(), d ** 2, atol=1e-5)
    assert float(out.dists[1, 40:].abs().max()) == 0 and int(out.idx[1, 40:].abs().max()) == 0   # padded queries
    assert torch.allclose(out.knn[0, :, 0], p1[0])
    short = knn_points(p1[:, :10], p1[1:2, :3].expand(2, 3, 3), K=5)            # K larger than the cloud
    assert float(short.dists[..., 3:].abs().max()) == 0
    assert knn_gather(p1, out.idx, lengths).shape == (2, 300, 6, 3)
    q = torch.rand(1, 20, 3, requires_grad=True)
    knn_points(q, torch.rand(1, 30, 3), K=2).dists.sum().backward()
    assert q.grad.abs().sum() > 0
    x, y = torch.rand(1, 50, 3), torch.rand(1, 60, 3)
    cd, _ = chamfer_distance(x, y)
    d = torch.cdist(x[0], y[0]) ** 2
    assert abs(float(cd) - float(d.min(1)[0].mean() + d.min(0)[0].mean())) < 1e-6
    assert float(chamfer_distance(Pointclouds(x), Pointclouds(x.clone()))[0]) < 1e-10


def test_packed_padded_round_trip_and_compositing_against_closed_forms():
    vals, first = torch.arange(14.0).reshape(7, 2), torch.tensor([0, 5])
    pad = packed_to_padded(vals, first, 5)
    assert pad.shape == (2, 5, 2) and torch.equal(pad[0], vals[:5]) and torch.equal(pad[1, :2], vals[5:]) and float(pad[1, 2:].sum()) == 0
    assert torch.equal(padded_to_packed(pad, first, 7), vals)
    assert torch.equal(padded_to_packed(packed_to_padded(vals[:, 0], first, 5), first, 7), vals[:, 0])
    idx = torch.tensor([[[[0]], [[2]], [[-1]]]])                      # (N=1, K=3, H=1, W=1)
    alphas = torch.tensor([[[[0.5]], [[0.25]], [[9.0]]]])
    feats = torch.tensor([[1.0, 10.0, 100.0], [2.0, 20.0, 200.0]])   # (C=2, P=3)
    ws = weighted_sum(idx, alphas, feats)
    assert torch.allclose(ws[0, :, 0, 0], torch.tensor([0.5 * 1 + 0.25 * 100, 0.5 * 2 + 0.25 * 200]))
    assert torch.allclose(norm_weighted_sum(idx, alphas, feats)[0, :, 0, 0], ws[0, :, 0, 0] / 0.75)


def test_ico_sphere_sampling_and_mesh_io(tmp_path):
    m = ico_sphere(4)
    v, f = m.verts_packed(), m.faces_packed()
    assert v.shape == (2562, 3) and f.shape == (5120, 3) and torch.allclose(v.norm(dim=1), torch.ones(2562), atol=1e-6)
    edges = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(1)[0].unique(dim=0)
    assert v.shape[0] - edges.shape[0] + f.shape[0] == 2                                  # closed genus-0 surface
    assert float((m.faces_normals_packed() * v[f].mean(1)).sum(1).min()) > 0              # outward orientation
    assert abs(float(m.faces_areas_packed().sum()) - 4 * math.pi) < 0.02
    m.scale_verts_(0.5)                                                                   # config.py:117
    torch.manual_seed(0)
    p, n = sample_points_from_meshes(m, 4000, return_normals=True)
    assert p.shape == (1, 4000, 3) and float((p.norm(dim=-1) - 0.5).abs().max()) < 2e-3
    assert float((torch.nn.functional.normalize(p, dim=-1) * n).sum(-1).min()) > 0.99
    assert float(p.mean(1).abs().max()) < 0.03                                            # area-uniform
    obj, ply = str(tmp_path / "m.obj"), str(tmp_path / "m.ply")
    save_obj(obj, m.verts_packed(), m.faces_packed())
    v2, faces2, _ = load_obj(obj)
    assert torch.allclose(v2, m.verts_packed(), atol=1e-6) and torch.equal(faces2.verts_idx, m.faces_packed())
    save_ply(ply, m.verts_packed(), m.faces_packed())
    v3, f3 = load_ply(ply)
    assert torch.allclose(v3, m.verts_packed(), atol=1e-5) and torch.equal(f3, m.faces_packed())
    bunny = os.path.join("/root/reference/example_data/pointclouds/bunny-8000.ply")
    if os.path.isfile(bunny):                                                             # binary little-endian PLY
        vb, _ = load_ply(bunny)
        z = np.load(os.path.join(ROOT, "tests", "golden", "clouds.npz"))
        assert np.allclose(vb.numpy(), z["bunny_points"], atol=1e-6)


def test_easydict_stand_in():
    from easydict import EasyDict
    d = EasyDict({"a": {"b": 1, "c": [{"d": 2}]}, "x": 3})
    assert d.a.b == 1 and d.a.c[0].d == 2 and d["x"] == 3 and d.get("nope") is None
    d.a.e = {"f": 4}
    assert d.a.e.f == 4 and isinstance(d.a, EasyDict) and isinstance(dict(**d)["a"], dict)
    with pytest.raises(AttributeError):
        d.missing


# ---- GPU: objects of the shim drive the drop-in, wired like config.create_renderer --------------------------------------------
def _get_class_from_string(cls_str):    # DSS/utils/__init__.py:68-73
    i = cls_str.rfind(".")
    return getattr(importlib.import_module(cls_str[:i]), cls_str[i + 1:])


@pytest.mark.gpu
def test_shim_objects_drive_the_drop_in_renderer_built_like_create_renderer():
    import oracle
    import scenes
    dev = torch.device("cuda:0")
    opt = {"renderer_type": "dss_amd.renderer.SurfaceSplattingRenderer", "raster_type": "dss_amd.rasterizer.SurfaceSplatting",
           "compositor_type": "dss_amd.renderer.NormWeightedCompositor",
           "raster_params": {"backface_culling": False, "Vrk_invariant": True, "Vrk_isotropic": False, "clip_pts_grad": -1.0, "cutoff_threshold": 1.0,
                             "depth_merging_threshold": 0.05, "image_size": 128, "points_per_pixel": 5,
                             "radii_backward_scaler": 5, "bin_size": None, "max_points_per_bin": None}}
    Renderer, Raster = _get_class_from_string(opt["renderer_type"]), _get_class_from_string(opt["raster_type"])
    Settings = _get_class_from_string(opt["raster_type"][: opt["raster_type"].rfind(".")] + ".PointsRasterizationSettings")
    renderer = Renderer(rasterizer=Raster(cameras=FoVPerspectiveCameras(), raster_settings=Settings(**opt["raster_params"])),
                        compositor=_get_class_from_string(opt["compositor_type"])()).to(dev)
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)[::4].copy()
    nrm = nrm[::4].copy()
    P = pts.shape[0]
    world = torch.from_numpy(pts).to(dev).requires_grad_(True)
    col = torch.rand(P, 3, device=dev, requires_grad=True)
    cloud = Pointclouds(world[None], normals=torch.from_numpy(nrm).to(dev)[None], features=col[None])
    R, T = look_at_view_transform((2.0, 2.4), (20.0, -10.0), (30.0, 200.0))
    cams = FoVPerspectiveCameras(R=R, T=T, znear=0.1, zfar=100.0, device=dev)
    rgba = renderer(cloud, cameras=cams)
    assert rgba.shape == (2, 128, 128, 4) and float(rgba[..., 3].detach().mean()) > 0.05
    g = torch.randn(rgba.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    (rgba * g).sum().backward()
    # the same two views through the oracle
    Mx = cams.get_full_projection_transform().get_matrix().cpu().numpy()
    Vx = cams.get_world_to_view_transform().get_matrix().cpu().numpy()
    h = scenes.global_h(pts)
    want_img, gp, gf = [], np.zeros((P, 3)), np.zeros((P, 3))
    for n in range(2):
        sc = scenes.setup_scene(pts, nrm, Mx[n:n + 1], Vx[n:n + 1], 128, h=h)
        idx, zb, qv, occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                                sc["num_pts"], 128, 5, 0.05)
        want_img.append(oracle.blend_forward(idx, qv, occ, sc["scaler"], col.detach().cpu().numpy())[0])
    want_img = np.stack(want_img)
    err = np.abs(rgba.detach().cpu().numpy() - want_img)
    assert err.max() <= 1e-4, (err.max(), err.mean(), float(renderer.rasterizer._Vrk_h.flatten()[0]), h)
    assert torch.isfinite(world.grad).all() and float(world.grad.abs().sum()) > 0 and float(col.grad.abs().sum()) > 0
    # self kNN of the shim on GPU tensors runs on the HIP grid search and equals the brute-force CPU route
    pp = torch.from_numpy(pts).to(dev)[None]
    a = knn_points(pp, pp, K=8)
    b = knn_points(pp.cpu(), pp.cpu(), K=8)
    assert torch.equal(a.idx.cpu(), b.idx) and torch.allclose(a.dists.cpu(), b.dists, atol=1e-6)
    # a padded batch of two clouds of different lengths (the padded rows go to the kernel as they lie: no host sync)
    half = pp.shape[1] // 2
    two = torch.stack([pp[0, :half], pp[0, half:2 * half]])
    ln = torch.tensor([half, half - 300], device=dev)
    a2 = knn_points(two, two, ln, ln, K=12)
    b2 = knn_points(two.cpu(), two.cpu(), ln.cpu(), ln.cpu(), K=12)
    assert torch.equal(a2.idx.cpu(), b2.idx) and torch.allclose(a2.dists.cpu(), b2.dists, atol=1e-6)
    assert float(a2.dists[1, half - 300:].abs().sum()) == 0 and int(a2.idx[1, half - 300:].abs().sum()) == 0
    # padded -> packed on the GPU (fixed-size nonzero, no host sync) == the CPU route
    xp = torch.rand(3, 50, 3, device=dev)
    fi = torch.tensor([0, 20, 70], device=dev)     # 20, 50 and 30 rows
    assert torch.equal(padded_to_packed(xp, fi, 100).cpu(), padded_to_packed(xp.cpu(), fi.cpu(), 100))
    assert torch.equal(padded_to_packed(xp[..., 0], fi, 100).cpu(), padded_to_packed(xp[..., 0].cpu(), fi.cpu(), 100))


def test_camera_center_closed_form_equals_the_inverse_of_the_world_to_view_transform():
    """`get_camera_center()` = last row of the inverse of the composed 4 x 4 world-to-view transform (pytorch3d's definition);
    the stand-in evaluates C = -T R^-1 in closed form (the reference's texture asks for one centre per POINT: cameras.py).
    Same value for rotations, for general invertible R, for per-call R / T overrides, and it carries gradients."""
    torch.manual_seed(3)
    R, T = look_at_view_transform([1.3, 2.0, 2.5], [10.0, 30.0, -20.0], [0.0, 45.0, 250.0])
    cams = FoVPerspectiveCameras(R=R, T=T)
    want = cams.get_world_to_view_transform().inverse().get_matrix()[:, 3, :3]
    assert torch.allclose(cams.get_camera_center(), want, atol=1e-6)
    R2 = (R + 0.2 * torch.rand_like(R)).requires_grad_(True)
    T2 = (T + torch.rand_like(T)).requires_grad_(True)
    got = cams.get_camera_center(R=R2, T=T2)
    want2 = cams.get_world_to_view_transform(R=R2, T=T2).inverse().get_matrix()[:, 3, :3]
    assert torch.allclose(got, want2, atol=1e-5)
    g_closed = torch.autograd.grad(got.sum(), (R2, T2))
    g_lu = torch.autograd.grad(want2.sum(), (R2, T2))
    assert all(torch.allclose(a, b, atol=1e-4) for a, b in zip(g_closed, g_lu))
    big = FoVPerspectiveCameras(R=R.repeat(1000, 1, 1), T=T.repeat(1000, 1))     # one camera per point, as the texture makes them
    assert torch.allclose(big.get_camera_center(), want.repeat(1000, 1), atol=1e-6)
