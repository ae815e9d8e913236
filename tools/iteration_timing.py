"""Time one whole training iteration of train_mvr.py on the HIP path (BASELINE configs[1] scene): render forward ->
image loss (Trainer.calc_dr_loss) -> render backward, and the projection regulariser with its fresh kNN-12, with the
image loss either fused (dss_amd.losses.calc_dr_loss) or written in torch ops.

    python tools/iteration_timing.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform  # noqa: E402
from dss_amd.cloud import PointClouds3D  # noqa: E402
from dss_amd.losses import ProjectionLoss, calc_dr_loss  # noqa: E402
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting  # noqa: E402
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer  # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    S = 512
    pts, nrm = scenes.load_cloud("bunny")
    pts, nrm = scenes.upsample_jitter(scenes.normalize_unit_sphere(pts), nrm, 4, seed=0)
    rng = np.random.default_rng(0)
    col = rng.uniform(0, 1, pts.shape).astype(np.float32)
    R, T = look_at_view_transform(2.0, 30.0, [45.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=5, bin_size=None,
                                     clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                        fused=True)
    normals = torch.from_numpy(nrm).to(DEV)
    with torch.no_grad():
        target = renderer(PointClouds3D([torch.from_numpy(pts).to(DEV)], [normals], [torch.from_numpy(col).to(DEV)]))
    t_img = target[..., :3].contiguous()
    t_mask = target[..., 3].contiguous()
    P = torch.nn.Parameter(torch.from_numpy(pts * 0.9).to(DEV))
    C = torch.nn.Parameter(torch.full((pts.shape[0], 3), 0.5, device=DEV))
    proj = ProjectionLoss(reduction="mean", filter_scale=2.0, knn_k=12)

    def torch_image_loss(img_pred):
        inside = t_mask.bool() & img_pred[..., 3].bool()
        rgb = (t_img - img_pred[..., :3]).abs()[inside].sum(-1).mean()
        p = img_pred[..., 3]
        iou = (1.0 - (p * t_mask).sum((1, 2)) / (p + t_mask - p * t_mask).sum((1, 2))).mean()
        return rgb + (t_mask - p).abs().mean() + 0.01 * iou

    def iteration(image_loss, regulariser):
        P.grad = C.grad = None
        pc = PointClouds3D([P], [normals], [C])
        loss = image_loss(renderer(pc))
        if regulariser:
            loss = loss + 0.01 * proj(pc, rebuild_knn=True)
        loss.backward()

    fused = lambda im: calc_dr_loss(im, t_img, t_mask, 1.0, 1.0)["loss"]  # noqa: E731
    out = {"points": int(pts.shape[0]), "image": S}
    out["render_plus_fused_image_loss_us"] = timed(lambda: iteration(fused, False))
    out["render_plus_torch_image_loss_us"] = timed(lambda: iteration(torch_image_loss, False))
    out["iteration_fused_with_projection_us"] = timed(lambda: iteration(fused, True))
    img_pred = renderer(PointClouds3D([P], [normals], [C])).detach()
    out["fused_image_loss_fwd_bwd_us"] = timed(lambda: calc_dr_loss(img_pred.requires_grad_(True), t_img, t_mask)["loss"].backward())
    out["torch_image_loss_fwd_bwd_us"] = timed(lambda: torch_image_loss(img_pred.requires_grad_(True)).backward())
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
