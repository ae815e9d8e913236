#!/usr/bin/env python3
"""Developer tool (GPU box): the bench step at BASELINE configs[1] with the cloud as it is and in Morton order
(dss_amd.cloud.spatial_order), ten steps per graph launch."""
import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch, bench
from dss_amd.cloud import spatial_order
dev = torch.device("cuda:0")
pts, nrm, col, h = bench.bunny_cloud()
for tag in ("as_is", "morton"):
    if tag == "morton":
        o = spatial_order(torch.from_numpy(pts)).numpy(); pts, nrm, col = pts[o].copy(), nrm[o].copy(), col[o].copy()
    wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0), cloud=(pts, nrm, col, h))
    for _ in range(5): wl.step()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): wl.step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(10): wl.step()
    best = 1e9
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
    print(tag, "ms/step", round(best, 5), "Msplats/s", round(wl.P / best / 1e3, 1))
