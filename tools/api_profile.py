#!/usr/bin/env python3
"""Developer tool (GPU box): where the HOST time of one iteration through the drop-in API goes
(SurfaceSplattingRenderer(fused)(cloud) + .backward() on the bench workload): cProfile, top functions by cumulative time."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dss_amd.cloud import PointClouds3D  # noqa: E402
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting  # noqa: E402
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
R, T = bench.look_at_view_transform(2.0, 30.0, [45.0])
cams = bench.FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R, T=T, device=dev)
st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=bench.CUTOFF, depth_merging_threshold=bench.THR,
                                 Vrk_invariant=True, Vrk_isotropic=False, radii_backward_scaler=bench.RADII_S,
                                 image_size=wl.S, points_per_pixel=bench.K, bin_size=None, clip_pts_grad=bench.CLIP,
                                 antialiasing_sigma=bench.SIGMA)
renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(), fused=True)
X = torch.nn.Parameter(wl.world.clone())
C = torch.nn.Parameter(wl.colors[:wl.Pc].clone())
h = wl.h[:1].clone()


def step():
    X.grad = None
    C.grad = None
    img = renderer(PointClouds3D([X], [wl.normals], [C]), Vrk_h=h)
    img.backward(wl.grad_out)


for _ in range(20):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print("ms per iteration: %.4f" % ((time.perf_counter() - t) / 200 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
