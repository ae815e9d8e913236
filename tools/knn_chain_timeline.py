#!/usr/bin/env python3
"""Developer tool (GPU box): per-kernel durations and gaps of the kNN-7 variance-scale chain (dss_knn_kth_sqdist +
dss_cloud_mean_clamp) at the metric's cloud (32,684 points), from a rocprofv3 kernel trace of 200 back-to-back calls.
    python tools/knn_chain_timeline.py [view]      (view: the per-camera culled search of the class path, 8 cameras)"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "knn_timeline")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from dss_amd import ops
    dev = torch.device("cuda:0")
    view = len(sys.argv) > 2 and sys.argv[2] == "view"
    wl = bench.Workload(dev, 8 if view else 1, bench.RowPartition(bench.S, 1, 0))
    one = torch.zeros(1, dtype=torch.int64, device=dev)
    cnt = torch.full((1,), wl.Pc, dtype=torch.int64, device=dev)
    for _ in range(220):
        if view:
            d = ops.knn_kth_sqdist_view(wl.world, one, cnt, 7, wl.V, wl.znear, wl.zfar, True, radius=0.2)
            ops.renderable_mean_clamp(d, wl.world, wl.V, wl.znear, wl.zfar, one.expand(8).contiguous(), cnt.expand(8).contiguous(), True,
                                      0.5, 5e-5, 1e-3, 0.5e-3, 7)
        else:
            ops.cloud_mean_clamp(ops.knn_kth_sqdist(wl.world, one, cnt, 7), one, cnt, 0.5, 5e-5, 1e-3, 0.5e-3, 7)
    torch.cuda.synchronize()
    sys.exit(0)
shutil.rmtree(OUT, ignore_errors=True)
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", OUT, "--", sys.executable, os.path.abspath(__file__),
                "--child"] + sys.argv[1:2], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
rows = []
for f in glob.glob(os.path.join(OUT, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dss::", "")[:40]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("knn_bbox_partial")]
chains = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])][-150:]
n = collections.Counter(len(c) for c in chains).most_common(1)[0][0]
chains = [c for c in chains if len(c) == n]
acc = collections.OrderedDict()
for c in chains:
    prev = None
    for st, en, nm in c:
        d = acc.setdefault(nm, [[], []])
        d[0].append((en - st) / 1e3)
        d[1].append(0.0 if prev is None else (st - prev) / 1e3)
        prev = en
tot = 0.0
for nm, (d, g) in acc.items():
    print("  %-42s dur %6.2f us  gap before %5.2f us" % (nm, sum(d) / len(d), sum(g) / len(g)))
    tot += sum(d) / len(d) + sum(g) / len(g)
print("  chain (first start to last end): %.2f us over %d chains (rocprofv3 inflates each kernel by ~2-4 us)" % (tot, len(chains)))
