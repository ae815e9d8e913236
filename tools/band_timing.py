#!/usr/bin/env python3
"""Developer tool (GPU box): per-rank COMPUTE time of the multi-GPU bench step, emulated on one GPU: G cameras,
rank r's row band, visibility = union over all bands (computed by rendering every band once, untimed).
No collectives: this is the part of the 8-GPU step that RCCL time is added to."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dss_amd import ops
from dss_amd.distributed import RowPartition

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else G // 2
dev = torch.device("cuda:0")
S, K = bench.S, bench.K
wl = bench.Workload(dev, G, RowPartition(S, 1, 0))   # G cameras, single-rank object (no process group needed)
part = RowPartition(S, G, rank)
fwd = lambda rows: ops.render_forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors,
                                      S, K, bench.CUTOFF, bench.THR, bench.SIGMA, False, True, rows=rows)
vis_all = torch.zeros(wl.P, dtype=torch.bool, device=dev)
for r in range(G):
    vis_all |= fwd(RowPartition(S, G, r).rows)["visible"]
g_band = part.slice(wl.grad_out).contiguous()
bucket = torch.empty(wl.P * 6, device=dev)
gf, gp = bucket[:wl.P * 3].view(wl.P, 3), bucket[wl.P * 3:].view(wl.P, 3)

def step():
    f = fwd(part.rows)
    ops.render_backward(g_band, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], vis_all, wl.first,
                        wl.num, bench.RADII_S, -1.0, image_size=S, rows=part.rows, out=(gf, gp))
    return ops.project_backward(wl.world, wl.M, wl.V, wl.first, wl.num, gp, f["valid"], True, clip=bench.CLIP)

for _ in range(10): step()
torch.cuda.synchronize(); t = time.perf_counter()
n = 100
for _ in range(n): step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / n * 1e3
print("G=%d rank %d: per-rank compute %.1f us/step (%d cameras x %d points, band rows %s, %d visible of %d)" % (
    G, rank, ms * 1e3, G, wl.Pc, part.rows, int(vis_all.sum()), wl.P))
