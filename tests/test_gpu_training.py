"""GPU: a miniature inverse-rendering loop through the drop-in API (the shape of train_mvr.py's
Trainer.compute_loss: masked L1 on RGB + L1 on the occupancy channel), checking that the gradients
the HIP backward produces actually drive the point cloud towards the target."""
import numpy as np
import pytest
import torch

import scenes
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
from dss_amd.cloud import PointClouds3D
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("fused", [False, True])
def test_inverse_rendering_loss_decreases(fused):
    torch.manual_seed(0)
    S, n_cams = 128, 4
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    col = (0.5 + 0.5 * nrm).astype(np.float32)
    R, T = look_at_view_transform(2.0, 25.0, [20.0 + 90.0 * k for k in range(n_cams)])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=5, bin_size=None,
                                     clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                        fused=fused)
    normals = torch.from_numpy(nrm).to(DEV)
    with torch.no_grad():
        target = renderer(PointClouds3D([torch.from_numpy(pts).to(DEV)], [normals], [torch.from_numpy(col).to(DEV)]))
    assert tuple(target.shape) == (n_cams, S, S, 4) and target[..., 3].mean() > 0.05

    # start from a shrunk, shifted and recoloured cloud
    P = torch.nn.Parameter(torch.from_numpy(pts * 0.85 + np.array([0.06, -0.04, 0.03], np.float32)).to(DEV))
    C = torch.nn.Parameter(torch.full((pts.shape[0], 3), 0.5, device=DEV))
    opt = torch.optim.Adam([{"params": [P], "lr": 2e-3}, {"params": [C], "lr": 2e-2}])
    losses, p_err = [], []
    for it in range(40):
        opt.zero_grad()
        img = renderer(PointClouds3D([P], [normals], [C]))
        mask = (target[..., 3:] * img[..., 3:]).detach()
        loss_rgb = ((img[..., :3] - target[..., :3]).abs() * mask).sum() / mask.sum().clamp_min(1) / 3
        loss_sil = (img[..., 3] - target[..., 3]).abs().mean()
        loss = loss_rgb + loss_sil
        loss.backward()
        assert torch.isfinite(P.grad).all() and torch.isfinite(C.grad).all()
        opt.step()
        losses.append(loss.item())
        p_err.append(float((P.detach() - torch.from_numpy(pts).to(DEV)).norm(dim=1).mean()))
    assert losses[-1] < 0.6 * losses[0], losses[::8]
    assert p_err[-1] < 0.9 * p_err[0], p_err[::8]   # geometry moved towards the target, not just colours


def test_iteration_with_fused_image_loss_and_projection_regulariser():
    """One train_mvr.py iteration end to end on the HIP path: render -> Trainer.calc_dr_loss (fused image loss) ->
    + 0.01 * ProjectionLoss with a fresh kNN-12 (configs/dss.yml:30-33, trainer.py:306-330) -> backward -> Adam.
    The fused image loss must give the same value and parameter gradients as the same loss written in torch ops."""
    from dss_amd.losses import ProjectionLoss, calc_dr_loss
    torch.manual_seed(0)
    S, n_cams = 128, 4
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    col = (0.5 + 0.5 * nrm).astype(np.float32)
    R, T = look_at_view_transform(2.0, 25.0, [20.0 + 90.0 * k for k in range(n_cams)])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=5, bin_size=None,
                                     clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor())
    normals = torch.from_numpy(nrm).to(DEV)
    with torch.no_grad():
        target = renderer(PointClouds3D([torch.from_numpy(pts).to(DEV)], [normals], [torch.from_numpy(col).to(DEV)]))
    img_nchw = target[..., :3].permute(0, 3, 1, 2).contiguous()   # the data loader's layout
    mask_img = target[..., 3:].permute(0, 3, 1, 2).contiguous()

    P = torch.nn.Parameter(torch.from_numpy(pts * 0.85 + np.array([0.06, -0.04, 0.03], np.float32)).to(DEV))
    C = torch.nn.Parameter(torch.full((pts.shape[0], 3), 0.5, device=DEV))

    # (a) same loss in plain torch ops (the arithmetic of trainer.py:332-372), for the gradient comparison
    def torch_loss(img_pred):
        t_img, t_mask = img_nchw.permute(0, 2, 3, 1), mask_img.reshape(-1, S, S)
        inside = t_mask.bool() & img_pred[..., 3].bool()
        rgb = (t_img - img_pred[..., :3]).abs()[inside].sum(-1).mean()
        p, t = img_pred[..., 3], t_mask
        iou = (1.0 - (p * t).sum((1, 2)) / (p + t - p * t).sum((1, 2))).mean()
        return rgb + (t - p).abs().mean() + 0.01 * iou

    pc = PointClouds3D([P], [normals], [C])
    img_pred = renderer(pc)
    ref = torch_loss(img_pred)
    ref.backward()
    gP, gC = P.grad.clone(), C.grad.clone()
    P.grad = C.grad = None
    fused = calc_dr_loss(renderer(PointClouds3D([P], [normals], [C])), img_nchw.permute(0, 2, 3, 1), mask_img, 1.0, 1.0)
    assert abs(fused["loss"].item() - ref.item()) <= 1e-5 * abs(ref.item())
    fused["loss"].backward()
    assert (P.grad - gP).norm() <= 1e-4 * gP.norm() and (C.grad - gC).norm() <= 1e-4 * gC.norm()

    # (b) the iteration proper, with the projection regulariser
    proj = ProjectionLoss(reduction="mean", filter_scale=2.0, knn_k=12)
    opt = torch.optim.Adam([{"params": [P], "lr": 2e-3}, {"params": [C], "lr": 2e-2}])
    hist = []
    for it in range(30):
        opt.zero_grad()
        pc = PointClouds3D([P], [normals], [C])
        loss = calc_dr_loss(renderer(pc), img_nchw.permute(0, 2, 3, 1), mask_img, 1.0, 1.0)
        total = loss["loss"] + 0.01 * proj(pc, rebuild_knn=True)
        total.backward()
        assert torch.isfinite(P.grad).all() and torch.isfinite(C.grad).all()
        opt.step()
        hist.append(total.item())
    assert hist[-1] < 0.7 * hist[0], hist[::6]
