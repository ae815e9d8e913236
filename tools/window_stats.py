#!/usr/bin/env python3
"""Developer tool (GPU box): sizes, in pixels, of the backward's two windows per visible point on a bench workload -- the splat's
own box (2 r + 1) and the occupancy window (2 rs + 1) -- i.e. how many of a task's lane slots the gather fills.
    python tools/window_stats.py cfg4|cfg5|cfg3|headline"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bench import K, CUTOFF, THR, SIGMA, RADII_S
from dss_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
dev = torch.device("cuda:0")
if which == "headline":
    wl = bench.Workload(dev, 2, bench.RowPartition(bench.S, 1, 0))
else:
    cloud, S_, N = bench.large_cloud(which)
    wl = bench.Workload(dev, N, bench.RowPartition(S_, 1, 0), cloud=cloud)
S = wl.S
f = ops.render_forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors, S, K, CUTOFF, THR,
                       SIGMA, False, True, rows=wl.part.rows, workspace_state=0)
vis = f["visible"].bool()
rs = ops.backward_radius(f["radii"], f["visible"], wl.first, wl.num, RADII_S)
r = f["radii"][vis].float().cpu().numpy() * S / 2     # NDC radius -> pixels
rs_px = rs.cpu().numpy() * S / 2
q = lambda a: [round(float(x), 2) for x in np.percentile(a, [5, 25, 50, 75, 95, 99])]
print(json.dumps({"workload": which, "S": S, "points": int(wl.P), "visible": int(vis.sum()),
                  "own_radius_px_x_pcts_5_25_50_75_95_99": q(r[:, 0]), "own_radius_px_y": q(r[:, 1]),
                  "search_radius_px_per_cloud": [round(float(x), 2) for x in rs_px],
                  "occupancy_window_columns": [int(2 * x) + 1 for x in rs_px]}))
