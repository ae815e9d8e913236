#!/usr/bin/env python3
"""Developer tool (GPU box): the drop-in API path (SurfaceSplattingRenderer forward + backward, eager) over the life of a
process -- host time per iteration and the GPU's own time for the same iteration's kernels (HIP events around each
iteration), in consecutive blocks of 100 iterations.  Shows whether a slow phase is the host or the GPU (clocks)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bench import *  # noqa: F401,F403
from dss_amd.cloud import PointClouds3D
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
dev = torch.device("cuda:0")
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
R, T = bench.look_at_view_transform(2.0, 30.0, [45.0])
cams = bench.FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R, T=T, device=dev)
st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=bench.CUTOFF, depth_merging_threshold=bench.THR,
                                 Vrk_invariant=True, Vrk_isotropic=False, radii_backward_scaler=bench.RADII_S,
                                 image_size=wl.S, points_per_pixel=bench.K, bin_size=None, clip_pts_grad=bench.CLIP,
                                 antialiasing_sigma=bench.SIGMA)
renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(), fused=True)
X = torch.nn.Parameter(wl.world.clone()); C = torch.nn.Parameter(wl.colors[:wl.Pc].clone()); h = wl.h[:1].clone()
def step():
    X.grad = None; C.grad = None
    img = renderer(PointClouds3D([X], [wl.normals], [C]), Vrk_h=h)
    img.backward(wl.grad_out)
for _ in range(3): step()
torch.cuda.synchronize()
rows = []
for blk in range(16):
    mt = (blk // 2) % 2 == 0   # two blocks with the autograd engine's device thread, two with the backward on the calling thread, ...
    torch.autograd.set_multithreading_enabled(mt)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for a, b in ev:
        a.record(); step(); b.record()
    torch.cuda.synchronize(); host = (time.perf_counter() - t) / 100 * 1e3
    gpu = sorted(a.elapsed_time(b) for a, b in ev)[50]
    rows.append({"block": blk, "engine_thread": mt, "host_ms_per_iteration": round(host, 4), "gpu_ms_first_to_last_kernel_median": round(gpu, 4)})
print(json.dumps(rows))
