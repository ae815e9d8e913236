#!/usr/bin/env python3
"""Developer tool: per-workgroup phase timestamps of the fine kernel (s_memtime) on the bench scene (or cfg3 / cfg4 / cfg5 of
tools/bench_large.py).  Builds a private -DDSS_FINE_TIMING copy of the library under gpurun_out/ and never touches the
shipped libdss_hip.so.  Usage (GPU box): python tools/fine_timing.py [cfg3|cfg4|cfg5|trained]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdss_hip_timing.so")
src = os.path.join(ROOT, "dss_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                *sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip")), "-o", so], check=True)
from dss_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
lib.dss_debug_set_fine_timing.argtypes = [ctypes.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if which == "cfg2":
    wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
    blocks = (bench.S // 8) ** 2  # DSS_TILE = 8
elif which == "trained":   # the clustered cloud of tools/clustered_timing.py: 8 cameras, 512^2, h from the kNN statistic (clamped)
    z = np.load(os.path.join(ROOT, "tests", "golden", "trained_cloud_cfg3.npz"))
    col = np.random.default_rng(0).uniform(0, 1, z["points"].shape).astype(np.float32)
    N, S = 8, 512
    wl = bench.Workload(dev, N, bench.RowPartition(S, 1, 0), cloud=(z["points"], z["normals"], col, None))
    blocks = N * (S // 8) ** 2
else:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes  # noqa: E402
    P, S, N = {"cfg4": (1_000_000, 1024, 8), "cfg5": (4_000_000, 2048, 1), "cfg3": (99_790, 512, 8)}[which]
    pts, nrm, col = scenes.synthetic_cloud(P, seed=0)
    h = scenes.global_h(pts[:: max(1, P // 200_000)]) * (200_000 / P if P > 200_000 else 1.0)
    h = float(np.clip(h, 5e-6, 1e-3))
    part = bench.RowPartition(S, 1, 0)
    if len(sys.argv) > 2:   # "row0,row1": the fine pass of that row band (a rank of the multi-GPU step)
        r0, r1 = (int(x) for x in sys.argv[2].split(","))
        part = bench.RowPartition(S, 3, 1, bounds=[0, r0, r1, S])
    wl = bench.Workload(dev, N, part, cloud=(pts, nrm, col, h))
    blocks = N * (S // 8) ** 2
buf = torch.zeros((2 * blocks + 4096, 12), dtype=torch.int64, device=dev)  # grid = queue slots (~tiles) + tiles/16 fill workgroups
for _ in range(3):
    wl.fine_kernel_ms(iters=5)
assert lib.dss_debug_set_fine_timing(ctypes.c_void_p(buf.data_ptr())) == 0
mean, med = wl.fine_kernel_ms(iters=20)
torch.cuda.synchronize()
t = buf.cpu().numpy()
print("fine kernel ms mean %.4f median %.4f (timing build)" % (mean, med))
cnt = t[:, 10]
busy = cnt > 0
fills = t[(t[:, 0] != 0) & (t[:, 10] == -1)]
if len(fills):
    f0 = fills[:, 8] - t[t[:, 0] != 0][:, 8].min()
    f1 = fills[:, 9] - t[t[:, 0] != 0][:, 8].min()
    print("fill workgroups %d: start p50 %d max %d | end p50 %d p90 %d max %d (ticks) | cycles mean %.0f max %d" % (
        len(fills), np.percentile(f0, 50), f0.max(), np.percentile(f1, 50), np.percentile(f1, 90), f1.max(),
        (fills[:, 7] - fills[:, 0]).mean(), (fills[:, 7] - fills[:, 0]).max()))
blk = np.arange(len(t))
sel = (t[:, 0] != 0) & (t[:, 10] != -1)
xcd = blk[sel] % 8   # observed placement: workgroup b runs on XCD b mod 8
print("tile workgroups per XCD:", np.bincount(xcd, minlength=8).tolist(), "| starting later than 3 us per XCD:",
      np.bincount(xcd[(t[sel, 8] - t[t[:, 0] != 0][:, 8].min()) > 300], minlength=8).tolist())
t = t[sel]  # workgroups that ran a tile (queue slots without work / flagged tiles exit before the first mark)
cnt = t[:, 10]
busy = cnt > 0
print("candidates per occupied tile: mean %.1f p50 %d p90 %d max %d | survivors of footprint 0: mean %.1f (%.2f of the candidates)" % (
    cnt[busy].mean(), np.percentile(cnt[busy], 50), np.percentile(cnt[busy], 90), cnt[busy].max(), t[busy, 11].mean(),
    t[busy, 11].sum() / max(cnt[busy].sum(), 1)))
rt0, rt1 = t[:, 8], t[:, 9]
print("realtime span (100MHz ticks): kernel %d, first start %d, last end %d" % (rt1.max() - rt0.min(), 0, rt1.max() - rt0.min()))
print("WG start spread (ticks): p50 %d p90 %d max %d" % tuple(np.percentile(rt0 - rt0.min(), [50, 90, 100])))
dur = t[:, 7] - t[:, 0]
print("tile workgroups %d | cycles mean %.0f p50 %.0f p90 %.0f max %d" % (
    int(busy.sum()), dur[busy].mean(), np.percentile(dur[busy], 50), np.percentile(dur[busy], 90), dur[busy].max()))
end = rt1 - rt0.min()
print("tile WG end time (ticks): p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(end[busy], [50, 90, 99, 100])))
late = busy & ((rt0 - rt0.min()) > 500)
print("tile WGs starting later than 5 us: %d, their mean count %.0f" % (int(late.sum()), cnt[late].mean() if late.any() else 0))
names = ["prologue(offset loads)", "stage chunk0", "cull chunk0", "survivors+rest chunks", "merge", "epilogue"]
for i, nm in enumerate(names):
    a, b = (i, i + 1) if i < 5 else (5, 7)
    d = (t[busy, b] - t[busy, a])
    print("  %-24s mean %8.0f  p90 %8.0f  max %8d cycles" % (nm, d.mean(), np.percentile(d, 90), d.max()))
heavy = np.argsort(-cnt)[:5]
for h in heavy:
    print("tile %4d count %4d:" % (h, cnt[h]), [int(t[h, k + 1] - t[h, k]) for k in range(5)], int(t[h, 7] - t[h, 5]),
          "start@%d end@%d" % (rt0[h] - rt0.min(), rt1[h] - rt0.min()))
np.save(os.path.join(out_dir, "fine_timing.npy"), t)
