#!/usr/bin/env python3
"""Developer tool: per-wavefront accounting of the fused persistent backward kernel (GPU box)."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdss_hip_timing.so"); src = os.path.join(ROOT, "dss_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-fno-fast-math", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                *sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip")),
                "-o", so], check=True)
from dss_amd import _lib, ops
_lib.LIB_PATH = so
import bench
dev = torch.device("cuda:0"); lib = _lib.load(); lib.dss_debug_set_occ_timing.argtypes = [ctypes.c_void_p]
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0)); S, K = bench.S, bench.K
info = ops.point_setup(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, S, 1.0, 1.0, False, True)
idx, zbuf, qv, occ, vis = ops.splat_points(info["pts_screen"], info["ellipse_params"], info["cutoff_threshold"], info["radii"],
                                           wl.first, wl.num, bench.THR, S, K, None, None, return_visible=True)
img, wsum = ops.blend_forward(idx, qv, occ, info["scaler"], wl.colors, return_wsum=True)
run = lambda: ops.render_backward(wl.grad_out, idx, qv, wsum, info["scaler"], info["pts_screen"], info["radii"], vis,
                                  wl.first, wl.num, bench.RADII_S, bench.CLIP)
for _ in range(5): run()
torch.cuda.synchronize()
buf = torch.zeros((max(wl.P, 16384), 6), dtype=torch.int64, device=dev)
assert lib.dss_debug_set_occ_timing(ctypes.c_void_p(buf.data_ptr())) == 0
run(); torch.cuda.synchronize()
t = buf.cpu().numpy(); t = t[t[:, 1] > 0]
print("waves recorded", len(t), "tasks total", int(t[:, 2].sum()), "tasks/wave max", int(t[:, 2].max()))
rt0 = t[:, 0].min()
print("wave start (100MHz ticks) p50 %d p90 %d max %d ; wave end p50 %d p90 %d max %d" % (
    *np.percentile(t[:, 0] - rt0, [50, 90, 100]), *np.percentile(t[:, 1] - rt0, [50, 90, 100])))
w = t[t[:, 2] > 0]
print("per task cycles: prologue %.0f occ %.0f blend+reduce %.0f" % (
    w[:, 5].sum() / w[:, 2].sum(), w[:, 3].sum() / w[:, 2].sum(), w[:, 4].sum() / w[:, 2].sum()))
print("wave busy ticks mean %.0f max %d" % ((w[:, 1] - w[:, 0]).mean(), (w[:, 1] - w[:, 0]).max()))
