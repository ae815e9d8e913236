#!/usr/bin/env python3
"""Developer tool (GPU box): per-workgroup phase stamps (100 MHz s_memrealtime) inside the single-launch backward
(render_backward_kernel<..., PREP>): when the segment / median workgroups finish, when the waiters are released, when the
launch ends.   python tools/fused_timing.py [mode 2|3]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdss_hip_timing.so"); src = os.path.join(ROOT, "dss_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-fno-fast-math", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                *[os.path.join(src, f) for f in ("api.hip", "raster_forward.hip", "raster_backward.hip", "blend.hip", "setup.hip",
                                                  "knn.hip", "shading.hip", "regularizers.hip", "image_loss.hip")],
                "-o", so], check=True)
from dss_amd import _lib, ops
_lib.LIB_PATH = so
import bench
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0"); lib = _lib.load(); lib.dss_debug_set_occ_timing.argtypes = [ctypes.c_void_p]
_lib.set_option(_lib.OPT_BACKWARD_FUSED, mode)
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0)); S, K = bench.S, bench.K
f = ops.render_forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors, S, K,
                       bench.CUTOFF, bench.THR, bench.SIGMA, False, True)
args = (wl.grad_out, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], f["visible"], wl.first, wl.num,
        bench.RADII_S, bench.CLIP)
run = lambda: ops.render_backward(*args, project=(wl.world, wl.M))
for _ in range(5): run()
torch.cuda.synchronize()
buf = torch.zeros((4096, 12), dtype=torch.int64, device=dev)
assert lib.dss_debug_set_occ_timing(ctypes.c_void_p(buf.data_ptr())) == 0
for rep in range(3):
    buf.zero_(); torch.cuda.synchronize()
    run(); torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64)
    used = t[:, 0] > 0
    t0 = t[used, 0].min()
    us = lambda a: (a - t0) / 100.0
    med = np.where(t[:, 4] > 0)[0]
    g = used.copy(); g[med] = False
    print("rep %d: %d workgroups (%d median); start spread %.2f us" % (rep, int(used.sum()), len(med), us(t[used, 0]).max()))
    for b in med[:2]:
        print("  median WG %d: start %.2f | inputs located %.2f | bucket chosen %.2f | candidates in %.2f | selected %.2f | published %.2f" % (
            b, us(t[b, 0]), us(t[b, 4]), us(t[b, 5]), us(t[b, 6]), us(t[b, 7]), us(t[b, 1])))
    a1 = us(t[g, 1]); rel = us(t[g, 2]); end = us(t[g, 3])
    print("  blend half done: min %.2f mean %.2f max %.2f us" % (a1.min(), a1.mean(), a1.max()))
    print("  released (rs seen): min %.2f mean %.2f max %.2f us" % (rel.min(), rel.mean(), rel.max()))
    print("  occupancy half done: min %.2f mean %.2f max %.2f us" % (end.min(), end.mean(), end.max()))
    if rep == 2:
        idx = np.where(g)[0]
        d2 = (t[g, 3] - t[g, 2]) / 100.0     # occupancy half per workgroup
        d1 = (t[g, 1] - t[g, 0]) / 100.0     # blend half per workgroup
        print("  occupancy half per workgroup: deciles", " ".join("%.2f" % np.percentile(d2, q) for q in range(0, 101, 10)))
        print("  blend half per workgroup    : deciles", " ".join("%.2f" % np.percentile(d1, q) for q in range(0, 101, 10)))
        for x in range(8):
            m = (idx % 8) == x
            print("    XCD %d: %4d workgroups, occupancy half mean %.2f max %.2f | blend half mean %.2f max %.2f" % (
                x, int(m.sum()), d2[m].mean(), d2[m].max(), d1[m].mean(), d1[m].max()))
        # by position in the grid (dispatch order): eighths of the workgroup index
        for o in range(8):
            m = (idx * 8 // (idx.max() + 1)) == o
            print("    grid eighth %d: occupancy half mean %.2f max %.2f | blend half mean %.2f" % (o, d2[m].mean(), d2[m].max(), d1[m].mean()))
        print("  correlation blend-half vs occupancy-half duration per workgroup: %.3f" % np.corrcoef(d1, d2)[0, 1])
