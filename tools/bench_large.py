#!/usr/bin/env python3
"""Developer tool (GPU box): the bench step on the larger BASELINE configs, single GPU.
  cfg4-like: synthetic 1M-point cloud, 8 ring cameras, 1024x1024
  cfg5-like: synthetic 4M-point cloud, 1 camera, 2048x2048
Prints ms/step (eager + graph) and the fine-kernel roofline numbers; meant to be run under rocprofv3."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
P, S, N = {"cfg4": (1_000_000, 1024, 8), "cfg5": (4_000_000, 2048, 1), "cfg3": (99_790, 512, 8)}[which]
pts, nrm, col = scenes.synthetic_cloud(P, seed=0)
h = scenes.large_cloud_h(pts)   # (the one definition shared with bench.py::large_cloud and tests/test_gpu_named_configs.py)
if os.environ.get("DSS_BENCH_MORTON") == "1":
    # spatially coherent point order (what a scanned or mesh-sampled cloud usually has; dss_amd.cloud.spatial_order)
    from dss_amd.cloud import spatial_order
    order = spatial_order(torch.from_numpy(pts)).numpy()
    pts, nrm, col = pts[order].copy(), nrm[order].copy(), col[order].copy()
if os.environ.get("BENCH_BACKWARD_TPW"):   # development A/B: DSS_OPT_BACKWARD_TPW
    from dss_amd import _lib
    _lib.set_option(_lib.OPT_BACKWARD_TPW, int(os.environ["BENCH_BACKWARD_TPW"]))
dev = torch.device("cuda:0")
wl = bench.Workload(dev, N, bench.RowPartition(S, 1, 0), cloud=(pts, nrm, col, h))
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 2 * wl.order_refresh if wl.order_refresh > 0 else 10   # (whole periods of the cached point order)
for _ in range(steps):
    wl.step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
fine_mean, fine_med = wl.fine_kernel_ms(iters=10)
K = bench.K
alg = wl.N * S * S * (12 * K + 4 + 16 + 4) + wl.P * 52
out = wl.step()
img = out[0]
rec = {"config": which, "order_refresh": wl.order_refresh, "point_order": "morton" if os.environ.get("DSS_BENCH_MORTON") == "1" else "random", "points_per_cloud": P, "cameras": N, "image_size": S, "ms_per_step_eager": round(ms, 4),
       "Msplats_per_s": round(wl.P / ms / 1e3, 2), "fine_kernel_ms": round(fine_mean, 4),
       "fine_algorithmic_bytes": alg, "fine_GBps": round(alg / fine_mean / 1e6, 1),
       "fine_frac_of_8TBps": round(alg / fine_mean / 1e6 / 8000, 4), "occupancy_mean": round(float(img[..., 3].mean()), 4), "h": h}
try:
    g_ms, b_ms, p_ms, pairs, n_vis = wl.backward_gather_ms(iters=10)
    rec.update(backward_gather_ms=round(g_ms, 4), backward_total_ms=round(b_ms, 4), backward_prep_ms=round(p_ms, 4),
               backward_pairs=pairs, backward_valu_frac=round(pairs * 12 / (g_ms * 1e-3) / 1e12 / (256 * 4 * 32 * 2.4e-3), 4))
except Exception as e:  # noqa: BLE001
    rec["backward_gather_error"] = str(e)[:200]
try:
    _f = bench.ops.render_forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors, S, K,
                                 bench.CUTOFF, bench.THR, bench.SIGMA, False, True)
    rec["visible_points"] = int(_f["visible"].sum())
    rec["median_radius_px"] = round(float(_f["radii"][_f["visible"]].median()) * S / 2, 2)
except Exception:
    pass
print(json.dumps(rec))
