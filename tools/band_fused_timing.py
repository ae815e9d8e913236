#!/usr/bin/env python3
"""Developer tool (GPU box): per-workgroup phase stamps (100 MHz s_memrealtime) inside the BAND variant of the two-phase
backward gather (render_backward_kernel<..., PREP, BAND>) for one emulated rank of the G-rank bench step: own-box filter,
blend half, release by the medians, search-radius filter, occupancy half.
    python tools/band_fused_timing.py [G] [layout cyclic|bands|balanced] [rank]"""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdss_hip_timing.so"); src = os.path.join(ROOT, "dss_amd", "csrc")
if not os.path.exists(so) or os.environ.get("TIMING_REBUILD", "1") == "1":
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                    "-fno-fast-math", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                    *[os.path.join(src, f) for f in ("api.hip", "raster_forward.hip", "raster_backward.hip", "blend.hip", "setup.hip",
                                                      "knn.hip", "shading.hip", "regularizers.hip", "image_loss.hip")],
                    "-o", so], check=True)
from dss_amd import _lib, ops
_lib.LIB_PATH = so
import bench
from dss_amd.distributed import RowPartition, balanced_bounds
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
layout = sys.argv[2] if len(sys.argv) > 2 else "cyclic"
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0"); lib = _lib.load(); lib.dss_debug_set_occ_timing.argtypes = [ctypes.c_void_p]
if os.environ.get("BAND_TPW"):
    _lib.set_option(_lib.OPT_BACKWARD_TPW, int(os.environ["BAND_TPW"]))
S, K = bench.S, bench.K
wl = bench.Workload(dev, G, RowPartition(S, 1, 0))
fwd = lambda rows: ops.render_forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors, S, K,
                                      bench.CUTOFF, bench.THR, bench.SIGMA, False, True, rows=rows)
bounds = None
if layout == "balanced":
    bounds = balanced_bounds(fwd(None)["occupancy"].sum(dim=(0, 2)).double().cpu(), G, align=8, min_rows=8)
parts = [RowPartition(S, G, r, cyclic=(layout == "cyclic"), bounds=bounds) for r in range(G)]
vis_all = torch.zeros(wl.P, dtype=torch.bool, device=dev)
for p in parts:
    vis_all |= fwd(p.rows)["visible"]
p = parts[rank]
f = fwd(p.rows)
g_band = p.slice(wl.grad_out).contiguous()
run = lambda: ops.render_backward(g_band, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], vis_all,
                                  wl.first, wl.num, bench.RADII_S, -1.0, image_size=S, rows=p.rows)
for _ in range(5): run()
torch.cuda.synchronize()
buf = torch.zeros((4096, 12), dtype=torch.int64, device=dev)
assert lib.dss_debug_set_occ_timing(ctypes.c_void_p(buf.data_ptr())) == 0
print("G %d layout %s rank %d rows %s: %d of %d points visible anywhere" % (G, layout, rank, p.rows, int(vis_all.sum()), wl.P))
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
    pub = us(t[med, 1])
    print("  medians published: min %.2f max %.2f" % (pub.min(), pub.max()))
    for name, col in (("own-box filter done", 8), ("blend half done", 1), ("released (rs seen)", 2), ("radius filter done", 9),
                      ("occupancy half done", 3)):
        a = us(t[g, col])
        print("  %-22s min %.2f mean %.2f p90 %.2f max %.2f us" % (name + ":", a.min(), a.mean(), np.percentile(a, 90), a.max()))
    if rep == 2:
        d2 = (t[g, 3] - t[g, 9]) / 100.0
        d1 = (t[g, 1] - t[g, 8]) / 100.0
        print("  occupancy half per workgroup: deciles", " ".join("%.2f" % np.percentile(d2, q) for q in range(0, 101, 10)))
        print("  blend half per workgroup    : deciles", " ".join("%.2f" % np.percentile(d1, q) for q in range(0, 101, 10)))
