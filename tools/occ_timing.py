#!/usr/bin/env python3
"""Developer tool: per-wavefront timestamps of occ_backward on the bench scene (GPU box)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdss_hip_timing.so")
src = os.path.join(ROOT, "dss_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                *[os.path.join(src, f) for f in ("api.hip", "raster_forward.hip", "raster_backward.hip", "blend.hip",
                                                  "setup.hip", "knn.hip")], "-o", so], check=True)
from dss_amd import _lib, ops  # noqa: E402
_lib.LIB_PATH = so
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
lib.dss_debug_set_occ_timing.argtypes = [ctypes.c_void_p]
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
S, K = bench.S, bench.K
info = ops.point_setup(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, S, 1.0, 1.0, False, True)
idx, zbuf, qv, occ, vis = ops.splat_points(info["pts_screen"], info["ellipse_params"], info["cutoff_threshold"],
                                           info["radii"], wl.first, wl.num, bench.THR, S, K, None, None,
                                           return_visible=True)
rs = ops.backward_radius(info["radii"], vis, wl.first, wl.num, bench.RADII_S)
gocc = wl.grad_out[..., 3]
buf = torch.zeros((wl.P, 6), dtype=torch.int64, device=dev)
for _ in range(5):
    ops.occ_backward(info["pts_screen"], info["radii"], vis, rs, gocc, wl.first, wl.num)
torch.cuda.synchronize()
assert lib.dss_debug_set_occ_timing(ctypes.c_void_p(buf.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.occ_backward(info["pts_screen"], info["radii"], vis, rs, gocc, wl.first, wl.num)
e1.record()
torch.cuda.synchronize()
t = buf.cpu().numpy()
v = vis.cpu().numpy()
print("kernel ms", e0.elapsed_time(e1), "rs px", float(rs[0]) * S / 2, "visible", int(v.sum()))
rt0, rt1 = t[:, 4], t[:, 5]
print("realtime span ticks(100MHz):", rt1.max() - rt0.min(), " wave start p50/p90/max:",
      np.percentile(rt0 - rt0.min(), [50, 90, 100]))
a = t[v]
print("active waves: prologue %.0f  loop %.0f  reduce %.0f cycles (mean); total mean %.0f p90 %.0f max %d" % (
    (a[:, 1] - a[:, 0]).mean(), (a[:, 2] - a[:, 1]).mean(), (a[:, 3] - a[:, 2]).mean(),
    (a[:, 3] - a[:, 0]).mean(), np.percentile(a[:, 3] - a[:, 0], 90), (a[:, 3] - a[:, 0]).max()))
i = t[~v]
print("inactive waves total mean %.0f cycles" % (i[:, 3] - i[:, 0]).mean())
# start time vs point index (dispatch order)
q = np.linspace(0, wl.P - 1, 9).astype(int)
print("start tick by point index:", [(int(k), int(rt0[k] - rt0.min())) for k in q])
