#!/usr/bin/env python3
"""Developer tool: phase timing (100 MHz realtime ticks) inside backward_prep_kernel (GPU box)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdss_hip_timing.so"); src = os.path.join(ROOT, "dss_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-fno-fast-math", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                *[os.path.join(src, f) for f in ("api.hip", "raster_forward.hip", "raster_backward.hip", "blend.hip", "setup.hip", "knn.hip")],
                "-o", so], check=True)
from dss_amd import _lib, ops
_lib.LIB_PATH = so
import bench
dev = torch.device("cuda:0"); lib = _lib.load(); lib.dss_debug_set_occ_timing.argtypes = [ctypes.c_void_p]
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0)); S, K = bench.S, bench.K
info = ops.point_setup(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, S, 1.0, 1.0, False, True)
idx, zbuf, qv, occ, vis = ops.splat_points(info["pts_screen"], info["ellipse_params"], info["cutoff_threshold"], info["radii"],
                                           wl.first, wl.num, bench.THR, S, K, None, None, return_visible=True)
print("P", wl.P, "visible", int(vis.sum()))
run = lambda: ops.backward_radius(info["radii"], vis, wl.first, wl.num, bench.RADII_S)
for _ in range(5): run()
torch.cuda.synchronize()
buf = torch.zeros((4096, 12), dtype=torch.int64, device=dev)
assert lib.dss_debug_set_occ_timing(ctypes.c_void_p(buf.data_ptr())) == 0
run(); torch.cuda.synchronize()
t = buf.cpu().numpy()
k1 = t[:(wl.P + 2047) // 2048]
print("compact WGs (us): mean %.2f max %.2f ; span first-start..last-end %.2f" % (
    ((k1[:, 1] - k1[:, 0]) / 100.0).mean(), ((k1[:, 1] - k1[:, 0]) / 100.0).max(), (k1[:, 1].max() - k1[:, 0].min()) / 100.0))
m = t[0, 6:12]
print("median WG (us): start after compact end %.2f | hist0 sum+select %.2f flatten+hist1 %.2f select1 %.2f hist2 %.2f select2 %.2f total %.2f" % (
    (m[0] - k1[:, 1].max()) / 100.0, *[(m[i + 1] - m[i]) / 100.0 for i in range(5)], (m[5] - m[0]) / 100.0))
