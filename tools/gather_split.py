#!/usr/bin/env python3
"""Developer tool (GPU box): how the backward gather's time splits between its two halves -- the occupancy window sweep
and the blend backward over the splat's own box -- by timing dss_render_backward_gather with and without grad_feat."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import scenes  # noqa: E402
from dss_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dev = torch.device("cuda:0")
if which == "cfg2":
    wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
    S = bench.S
else:
    P, S, N = {"cfg4": (1_000_000, 1024, 8), "cfg5": (4_000_000, 2048, 1), "cfg3": (99_790, 512, 8)}[which]
    pts, nrm, col = scenes.synthetic_cloud(P, seed=0)
    h = scenes.global_h(pts[:: max(1, P // 200_000)]) * (200_000 / P if P > 200_000 else 1.0)
    wl = bench.Workload(dev, N, bench.RowPartition(S, 1, 0), cloud=(pts, nrm, col, float(np.clip(h, 5e-6, 1e-3))))
f = ops.render_forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors, S, bench.K,
                       bench.CUTOFF, bench.THR, bench.SIGMA, False, True)
args = (wl.grad_out, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], f["visible"], wl.first, wl.num,
        bench.RADII_S, bench.CLIP)
out = {"config": which}
for feats in (True, False):
    gf, gp, rs = ops.render_backward(*args, with_features=feats, return_rs=True)
    if gf is None:
        gf = torch.empty((wl.P, 3), device=dev)
    o = (gf, gp)
    full = lambda: ops.render_backward(*args, with_features=feats, out=o if feats else None)
    if feats:
        gather = lambda: ops.render_backward(*args, out=o, gather_only_rs=rs)
        full()
        out["gather_with_blend_ms"] = bench.Workload._event_ms(gather, 30)[0]
    out["full_%s_ms" % ("with_blend" if feats else "occupancy_only")] = bench.Workload._event_ms(full, 30)[0]
print(json.dumps(out))
