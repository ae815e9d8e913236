#!/usr/bin/env python3
"""Developer tool (GPU box, torchrun with 2 ranks on one GPU, gloo): where the two-rank step differs from the single-rank one."""
import os, sys, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("gloo")
ref = bench.Workload(dev, world, bench.RowPartition(bench.S, 1, 0))
img1, gw1, gc1 = ref.step()
img1 = img1.clone()
rel = lambda a, b: float((a - b).norm() / b.norm())
order = [bool(int(c)) for c in os.environ.get("DBG_ORDER", "01")]
for cyclic in order:
    wl = bench.Workload(dev, world, bench.RowPartition(bench.S, world, rank, cyclic=cyclic))
    print("rank", rank, "cyclic", cyclic, "h", getattr(wl, "h", None), "ref h", getattr(ref, "h", None), flush=True)
    for it in range(3):
        img, gw, gc = wl.step()
        torch.cuda.synchronize()
        d = (img != img1)
        rows = d.flatten(2).any(2) if d.dim() == 4 else d
        bad = torch.nonzero(d.reshape(d.shape[0], d.shape[1], -1).any(2))
        print("rank", rank, "cyclic", cyclic, "step", it, "differing pixels", int(d.sum()), "rows",
              sorted(set(bad[:, 1].tolist()))[:40], "cams", sorted(set(bad[:, 0].tolist())),
              "grad rel", rel(gw, gw1), rel(gc, gc1), flush=True)
dist.destroy_process_group()
