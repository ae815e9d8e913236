#!/usr/bin/env python3
"""Developer tool (GPU box): the library's kernels on a CLUSTERED cloud -- the model of the reference's own train_mvr.py at
BASELINE configs[2] after 893 iterations on the HIP kernels (tests/golden/trained_cloud_cfg3.npz, written from the
checkpoint of tools/train_mvr_ref.py): a dense bulk (median radius 0.33, 7th-neighbour distance 0.002-0.01) inside a thin
halo out to radius 2.1 (7th-neighbour distance 0.1-0.3), hundreds of overlapping splats per pixel.  The uniform clouds
of bench.py do not reach this state; the training loop lives in it (profiles/r6_b_train_mvr_ref_kernel_stats.csv: fine
pass 0.2 ms at the start of the run, 2.6 ms at its end).
    python tools/clustered_timing.py [lib.so|default[@knn=3] ...]   # every library given (default: the shipped one), in order;
                                                                    # @knn=3: without the kNN skip structure (DSS_OPT_KNN_QUERY)
Prints one JSON line per library: the render step (8 cameras, 512^2, forward + backward), the fine pass alone, the
per-camera variance-scale search (kNN-7, fixed radius 0.2) and the kNN-12 search with indices of the regularisers."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
from dss_amd import _lib
if %(so)r: _lib.LIB_PATH = %(so)r
import bench
from dss_amd import ops
if %(knn)r: _lib.set_option(_lib.OPT_KNN_QUERY, int(%(knn)r))
dev = torch.device("cuda:0")
z = np.load(os.path.join(%(root)r, "tests", "golden", "trained_cloud_cfg3.npz"))
pts, nrm = z["points"], z["normals"]
col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
N, S = 8, 512
wl = bench.Workload(dev, N, bench.RowPartition(S, 1, 0), cloud=(pts, nrm, col, None))
def timed(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best
step = timed(wl.step, 20)
fine = min(wl.fine_kernel_ms(iters=20)[0] for _ in range(2))
one = torch.zeros(1, dtype=torch.int64, device=dev); cnt = torch.full((1,), wl.Pc, dtype=torch.int64, device=dev)
# the class path's variance-scale search: cameras of the training distance with the near plane inside the halo (some cull)
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
R, T = look_at_view_transform(1.7, 30.0, [45.0 * k for k in range(N)])
cam = FoVPerspectiveCameras(znear=1.0, zfar=100.0, fov=60.0, R=R, T=T)
V = cam.get_world_to_view_transform().get_matrix().to(dev).contiguous()
zn, zf = torch.full((N,), 1.0, device=dev), torch.full((N,), 100.0, device=dev)
first = torch.zeros(1, dtype=torch.int64, device=dev)
knn_view = timed(lambda: ops.knn_kth_sqdist_view(wl.world, first, cnt, 7, V, zn, zf, True, 0.2), 10)
knn_plain = timed(lambda: ops.knn_kth_sqdist(wl.world, one, cnt, 7), 10)
knn12 = timed(lambda: ops.knn_points(wl.world, one, cnt, 12), 10)
print(json.dumps({"lib": (%(so)r or "shipped") + ("@knn=" + %(knn)r if %(knn)r else ""), "h": float(wl.h[0]), "step_ms": round(step, 4), "fine_ms": round(fine, 4),
                  "knn7_view_ms": round(knn_view, 4), "knn7_plain_ms": round(knn_plain, 4), "knn12_idx_ms": round(knn12, 4)}))
'''
for spec in (sys.argv[1:] or [""]):
    so, _, knn = spec.partition("@knn=")   # "default@knn=3": DSS_OPT_KNN_QUERY = 3, the uniform-grid walk without the skip structure
    so = os.path.join(ROOT, so) if so and so != "default" else ""
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "so": so, "knn": knn}], capture_output=True, text=True, timeout=900)
    print(r.stdout.strip() or ("FAILED " + r.stderr[-1500:]), flush=True)
