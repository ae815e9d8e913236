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
