#!/usr/bin/env python3
"""Developer tool: per-workgroup phase timestamps of the fine kernel (s_memtime) on the bench scene.
Builds a private -DDSS_FINE_TIMING copy of the library under gpurun_out/ and never touches the
shipped libdss_hip.so.  Usage (GPU box): python tools/fine_timing.py"""
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
                "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                *sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip")), "-o", so], check=True)
from dss_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
import bench  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
lib.dss_debug_set_fine_timing.argtypes = [ctypes.c_void_p]
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
blocks = (bench.S // 8) ** 2  # DSS_TILE = 8
buf = torch.zeros((blocks + 2048, 12), dtype=torch.int64, device=dev)  # grid = DSS_HEAVY_MAX + tiles workgroups
for _ in range(3):
    wl.fine_kernel_ms(iters=5)
assert lib.dss_debug_set_fine_timing(ctypes.c_void_p(buf.data_ptr())) == 0
mean, med = wl.fine_kernel_ms(iters=20)
torch.cuda.synchronize()
t = buf.cpu().numpy()
print("fine kernel ms mean %.4f median %.4f" % (mean, med))
cnt = t[:, 10]
busy = cnt > 0
t = t[t[:, 0] != 0]  # workgroups that ran a tile (queue slots without work / flagged tiles exit before the first mark)
cnt = t[:, 10]
busy = cnt > 0
rt0, rt1 = t[:, 8], t[:, 9]
print("realtime span (100MHz ticks): kernel %d, first start %d, last end %d" % (rt1.max() - rt0.min(), 0, rt1.max() - rt0.min()))
print("WG start spread (ticks): p50 %d p90 %d max %d" % tuple(np.percentile(rt0 - rt0.min(), [50, 90, 100])))
dur = t[:, 7] - t[:, 0]
print("WG cycles: empty tiles mean %.0f max %d | occupied mean %.0f p90 %.0f max %d" % (
    dur[~busy].mean(), dur[~busy].max(), dur[busy].mean(), np.percentile(dur[busy], 90), dur[busy].max()))
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
