#!/usr/bin/env python3
"""Developer tool (GPU box): the training loop of train_mvr.py at BASELINE configs[2] scale, entirely on the HIP path.

  target   bunny-8000 x12 tangent-plane jitter = 98,052 points (the box has no yoga6 scan; same size class as the
           99,790-point cloud of the config), 128 cameras drawn by the CameraSampler rule (DSS/core/camera.py:41-51),
           512^2 targets rendered with tri-colour point lights (common.py:47-89), batches of 8 views (dss.yml:27)
  model    dss_amd.model.Model: a sphere of as many points, learnable positions and colours
  step     model forward (shade, render, in-mask filter) -> Trainer.calc_dr_loss (fused) + 0.01 * ProjectionLoss with a
           fresh kNN-12 (dss.yml:30) -> backward -> Adam; radii_backward_scaler decays by 0.99 per iteration
           (scheduler.py:36-48)

Prints one JSON line: ms per iteration (wall clock, synchronised at both ends), splats per second, loss first/last.
    python tools/train_mvr_cfg3.py [iterations]
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from dss_amd.cameras import CameraSampler, FoVPerspectiveCameras  # noqa: E402
from dss_amd.cloud import PointClouds3D  # noqa: E402
from dss_amd.losses import ProjectionLoss, calc_dr_loss  # noqa: E402
from dss_amd.model import Model  # noqa: E402
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting  # noqa: E402
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer  # noqa: E402
from dss_amd.texture import LightingTexture, PointLights  # noqa: E402

DEV = "cuda:0"
S, BATCH, N_CAMS = 512, 8, 128


def sample_cameras():
    """128 cameras by the reference's CameraSampler rule (DSS/core/camera.py:41-51), distances in [1.2, 2.2]."""
    torch.manual_seed(0)
    sampler = CameraSampler(N_CAMS, BATCH, distance_range=[[1.2, 2.2]], sort_distance=True)
    return sampler.R, sampler.T


def tri_colour_lights(cams, g):
    """get_tri_color_lights_for_view(point_lights=True): blue / green / red lights on a half dome that is rotated so
    that its pole points at the camera (about a random roll), 5 units away."""
    elev = torch.tensor([30.0, 30.0, 30.0]) * math.pi / 180
    azim = torch.tensor([-60.0, 60.0, 180.0]) * math.pi / 180
    dome = torch.stack([torch.cos(elev) * torch.sin(azim), torch.sin(elev), torch.cos(elev) * torch.cos(azim)], -1)  # (3,3)
    cam_pos = cams.get_camera_center().cpu()
    up = torch.nn.functional.normalize(cam_pos, dim=-1)                                  # dome pole -> camera
    fwd = torch.nn.functional.normalize(torch.cross(cam_pos, torch.rand(cam_pos.shape, generator=g), dim=-1), dim=-1)
    side = torch.nn.functional.normalize(torch.cross(up, fwd, dim=-1), dim=-1)
    basis = torch.stack([side, up, fwd], dim=1)                                          # rows = dome x, y, z axes
    loc = torch.einsum("lk,nkj->nlj", dome, basis) * 5.0                                 # (N,3 lights,3)
    n = cam_pos.shape[0]
    return PointLights(ambient_color=torch.full((n, 1, 3), 0.2),
                       diffuse_color=torch.tensor([[0.0, 0.0, 0.8], [0.0, 0.8, 0.0], [0.8, 0.0, 0.0]]).expand(n, 3, 3),
                       specular_color=torch.zeros(n, 3, 3), location=loc, device=DEV)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    g = torch.Generator().manual_seed(0)
    pts, nrm = scenes.load_cloud("bunny")
    pts, nrm = scenes.upsample_jitter(scenes.normalize_unit_sphere(pts) * 0.5, nrm, 12, seed=0)
    P = pts.shape[0]
    R, T = sample_cameras()
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5.0, image_size=S, points_per_pixel=5, bin_size=None,
                                     clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(raster_settings=st), NormWeightedCompositor(), fused=True)
    batches = []
    tgt = PointClouds3D([torch.from_numpy(pts).to(DEV)], [torch.from_numpy(nrm).to(DEV)],
                        [torch.ones(P, 3, device=DEV)])
    texture = LightingTexture(device=DEV)
    with torch.no_grad():
        for b in range(N_CAMS // BATCH):
            cams = FoVPerspectiveCameras(znear=0.1, R=R[b * BATCH:(b + 1) * BATCH], T=T[b * BATCH:(b + 1) * BATCH], device=DEV)
            lights = tri_colour_lights(cams, g)
            rgba = renderer(texture(tgt, cameras=cams, lights=lights), cameras=cams)
            batches.append((cams, lights, rgba[..., :3].permute(0, 3, 1, 2).contiguous(), rgba[..., 3:].permute(0, 3, 1, 2).contiguous()))

    v = torch.randn(P, 3, generator=g)
    v = torch.nn.functional.normalize(v, dim=-1)
    model = Model((v * 0.45)[None], v[None], torch.ones(1, P, 3), renderer, texture=texture, device=DEV)
    proj = ProjectionLoss(reduction="mean", filter_scale=2.0, knn_k=12)
    opt = torch.optim.Adam([{"params": [model.points], "lr": 1e-3}, {"params": [model.normals], "lr": 1e-3},
                            {"params": [model.colors], "lr": 1e-2}])

    def iteration(it):
        cams, lights, img, mask_img = batches[it % len(batches)]
        opt.zero_grad()
        out = model(mask_img=mask_img, cameras=cams, lights=lights)
        loss = calc_dr_loss(out["rgba_pred"], img.permute(0, 2, 3, 1), mask_img, 1.0, 1.0)["loss"]
        loss = loss + 0.01 * proj(out["iso_pcl"], rebuild_knn=True, points_filter=model.points_filter)
        loss.backward()
        opt.step()
        st.radii_backward_scaler = max(1.0, st.radii_backward_scaler * 0.99)
        return loss

    first = [iteration(i).item() for i in range(len(batches))][: len(batches)]      # one epoch of warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(iters):
        last = iteration(len(batches) + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    tail = [iteration(len(batches) + iters + i).item() for i in range(len(batches))]
    print(json.dumps({"config": "cfg3-like train_mvr loop", "points": P, "views_per_iteration": BATCH, "image": S,
                      "cameras_total": N_CAMS, "iterations_timed": iters, "ms_per_iteration": round(ms, 3),
                      "Msplats_per_s": round(P * BATCH / ms / 1e3, 1), "loss_first_epoch_mean": round(float(np.mean(first)), 4),
                      "loss_last_epoch_mean": round(float(np.mean(tail)), 4)}))


if __name__ == "__main__":
    main()
