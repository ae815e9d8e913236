#!/usr/bin/env python3
"""Developer tool (GPU box): the two query kernels of the grid kNN (cooperative, 16 lanes per query / one thread per query,
DSS_OPT_KNN_QUERY) over the input size, for the K-th distance statistic (K = 7) and the K = 12 neighbour lists: event-timed
whole calls (grid build included), synthetic surface cloud (tests/scenes.py).  -> one JSON line"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, scenes  # noqa: E402
from dss_amd import _lib, ops  # noqa: E402
dev = torch.device("cuda:0")
out = []
for P in (32768, 65536, 131072, 262144, 524288, 1000000, 4000000):
    pts, _, _ = scenes.synthetic_cloud(P, seed=1)
    w = torch.from_numpy(pts).to(dev)
    one = torch.zeros(1, dtype=torch.int64, device=dev); cnt = torch.full((1,), P, dtype=torch.int64, device=dev)
    row = {"points": P}
    ref = None
    for name, opt in (("cooperative", 1), ("thread_per_query", 2)):
        _lib.set_option(_lib.OPT_KNN_QUERY, opt)
        kth = ops.knn_kth_sqdist(w, one, cnt, 7)
        if ref is None: ref = kth
        else: assert torch.equal(ref, kth), "the two kernels disagree"
        n = 20 if P <= 262144 else 6
        row["kth7_ms_" + name] = round(bench.Workload._event_ms(lambda: ops.knn_kth_sqdist(w, one, cnt, 7), n)[0], 4)
        row["knn12_ms_" + name] = round(bench.Workload._event_ms(lambda: ops.knn_points(w, one, cnt, 12), n)[0], 4)
    _lib.set_option(_lib.OPT_KNN_QUERY, 0)
    out.append(row)
print(json.dumps(out))
