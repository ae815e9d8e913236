#!/usr/bin/env python3
"""Developer tool (GPU box): the kNN-7 statistic chain (dss_knn_kth_sqdist + dss_cloud_mean_clamp) and the kNN-12 lists on the
bench cloud (32,684 points) and a 99,790-point cloud: event-timed totals; run under rocprofv3 --kernel-trace --stats for the
per-kernel split."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import scenes  # noqa: E402
from dss_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for name, reps in (("32k", 4), ("100k", 12)):
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    pts, _ = scenes.upsample_jitter(pts, nrm, reps, seed=0)
    w = torch.from_numpy(pts).to(dev)
    one = torch.zeros(1, dtype=torch.int64, device=dev)
    cnt = torch.full((1,), w.shape[0], dtype=torch.int64, device=dev)
    kth = lambda: ops.cloud_mean_clamp(ops.knn_kth_sqdist(w, one, cnt, 7), one, cnt, 0.5, 5e-5, 1e-3, 0.5e-3, 7)
    lists = lambda: ops.knn_points(w, one, cnt, 12)
    out[name] = {"points": int(w.shape[0]), "kth7_chain_ms": bench.Workload._event_ms(kth, 30)[0],
                 "knn12_lists_ms": bench.Workload._event_ms(lists, 30)[0]}
print(json.dumps(out))
