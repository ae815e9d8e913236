"""GPU: the point model of train_mvr.py (DSS/models/point_modeling.py) on the HIP path -- the in-mask filter against
the reference golden vectors and the oracle (flags: exact), Model.forward against a hand-composed render, and one
whole Trainer iteration (model -> calc_dr_loss -> projection regulariser with the model's filter)."""
import os

import numpy as np
import pytest
import torch

import oracle
import scenes
from dss_amd import ops
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
from dss_amd.cloud import PointClouds3D
from dss_amd.losses import ProjectionLoss, calc_dr_loss
from dss_amd.model import Model
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
from dss_amd.texture import LightingTexture, PointLights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inmask.npz")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_inmask_matches_oracle_exactly_and_the_reference_golden():
    z = np.load(GOLD)
    got = ops.points_inmask(_t(z["points"]), _t(z["M"]), _t(z["mask"]), _t(z["visibility"])).cpu().numpy()
    assert np.array_equal(got, oracle.points_inmask(z["points"], z["M"], z["mask"], z["visibility"]).astype(bool))
    assert (got != z["inmask"]).mean() <= 1e-3          # reference projection = batched matmul, see the pinning test
    every = ops.points_inmask(_t(z["points"]), _t(z["M"]), _t(z["mask"])[:, None]).cpu().numpy()   # (N,1,H,W), no flags
    assert np.array_equal(every, oracle.points_inmask(z["points"], z["M"], z["mask"]).astype(bool))
    # larger, non-square, points outside the frustum and behind the camera
    rng = np.random.default_rng(4)
    pts = rng.normal(0, 1.2, (200000, 3)).astype(np.float32)
    mask = (rng.random((3, 200, 333)) < 0.5).astype(np.float32)
    got = ops.points_inmask(_t(pts), _t(z["M"]), _t(mask)).cpu().numpy()
    assert np.array_equal(got, oracle.points_inmask(pts, z["M"], mask).astype(bool))
    with pytest.raises(RuntimeError, match="views"):
        ops.points_inmask(_t(pts), _t(z["M"]), _t(mask[:2]))


def _scene(S=96, n_cams=3):
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    R, T = look_at_view_transform(2.0, 25.0, [20.0 + 360.0 / n_cams * k for k in range(n_cams)])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=5, bin_size=None,
                                     clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor())
    return pts.astype(np.float32), nrm.astype(np.float32), cams, renderer


def test_model_forward_matches_hand_composed_render_and_fills_the_filter():
    pts, nrm, cams, renderer = _scene()
    P = pts.shape[0]
    col = np.random.default_rng(0).uniform(0.2, 1, pts.shape).astype(np.float32)
    lights = PointLights(location=((2.0, 2.0, -2.0),), device=DEV)
    texture = LightingTexture(device=DEV, cameras=cams, lights=lights)
    model = Model(_t(pts)[None], _t(nrm * 3.0)[None], _t(col)[None], renderer, texture=texture, device=DEV)
    act = torch.from_numpy(np.random.default_rng(1).random(P) < 0.8).to(DEV)
    model.points_activation.copy_(act[None])
    yy, xx = np.mgrid[0:96, 0:96]
    mask_img = _t(np.stack([((yy - 48) ** 2 + (xx - 40 - 8 * n) ** 2 < 30 ** 2) for n in range(3)]).astype(np.float32))[:, None]
    out = model(mask_img=mask_img, cameras=cams, lights=lights)

    # the same image composed by hand from the active points
    n_unit = torch.nn.functional.normalize(_t(nrm * 3.0), dim=-1)
    shaded = texture(PointClouds3D([_t(pts)[act]], [n_unit[act]], [_t(col)[act]]), cameras=cams, lights=lights)
    ref = renderer(shaded, cameras=cams)
    assert torch.equal(out["img_pred"], ref[..., :3]) and torch.equal(out["mask_img_pred"], ref[..., 3:])
    iso = out["iso_pcl"]
    Pf = int(act.sum())
    assert len(iso) == 1 and iso.points_packed().shape[0] == Pf
    assert torch.allclose(iso.normals_packed().norm(dim=1), torch.ones(Pf, device=DEV), atol=1e-5)
    flt = model.points_filter
    assert tuple(flt.activation.shape) == (1, P) and tuple(flt.visibility.shape) == (1, Pf) == tuple(flt.inmask.shape)
    assert flt.visibility.any() and not flt.visibility.all()
    M = cams.get_full_projection_transform().get_matrix().cpu().numpy()
    want = oracle.points_inmask(pts[act.cpu().numpy()], M, mask_img[:, 0].cpu().numpy(), flt.visibility[0].cpu().numpy())
    assert np.array_equal(flt.inmask[0].cpu().numpy(), want.astype(bool))
    assert not (flt.inmask & ~flt.visibility).any()
    # gradients reach all three parameter sets through shading + render
    (out["img_pred"].sum() + out["mask_img_pred"].sum()).backward()
    for prm in (model.points, model.normals, model.colors):
        assert prm.grad is not None and torch.isfinite(prm.grad).all() and prm.grad.abs().sum() > 0
    assert not model.points.grad[0][~act].any()        # inactive points get no gradient


def test_trainer_iteration_through_the_model():
    """trainer.py:290-330 with the HIP pieces: model forward -> calc_dr_loss -> 0.01 * ProjectionLoss(points_filter)."""
    pts, nrm, cams, renderer = _scene(S=128, n_cams=4)
    rng = np.random.default_rng(0)
    col = (0.5 + 0.5 * nrm).astype(np.float32)
    with torch.no_grad():
        target = renderer(PointClouds3D([_t(pts)], [_t(nrm)], [_t(col)]), cameras=cams)
    img = target[..., :3].permute(0, 3, 1, 2).contiguous()
    mask_img = target[..., 3:].permute(0, 3, 1, 2).contiguous()
    start = pts * 0.85 + np.array([0.06, -0.04, 0.03], np.float32)
    model = Model(_t(start)[None], _t(nrm)[None], torch.full((1, pts.shape[0], 3), 0.5), renderer, device=DEV)
    proj = ProjectionLoss(reduction="mean", filter_scale=2.0, knn_k=12)
    opt = torch.optim.Adam([{"params": [model.points], "lr": 2e-3}, {"params": [model.colors], "lr": 2e-2}])
    hist = []
    for it in range(30):
        opt.zero_grad()
        out = model(mask_img=mask_img, cameras=cams)
        rgba = torch.cat([out["img_pred"], out["mask_img_pred"]], dim=-1)
        loss = calc_dr_loss(rgba, img.permute(0, 2, 3, 1), mask_img, 1.0, 1.0)
        total = loss["loss"] + 0.01 * proj(out["iso_pcl"], rebuild_knn=True, points_filter=model.points_filter)
        total.backward()
        opt.step()
        hist.append(total.item())
    assert hist[-1] < 0.7 * hist[0], hist[::6]
    assert model.points_filter.inmask.shape == (1, pts.shape[0]) and model.points_filter.inmask.any()
