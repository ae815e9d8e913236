#!/usr/bin/env python3
"""Developer tool: A/B the XCD-aware tile mapping of the fine kernel (two private builds)."""
import os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "dss_amd", "csrc"); out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
files = [os.path.join(src, f) for f in ("api.hip", "raster_forward.hip", "raster_backward.hip", "blend.hip", "setup.hip", "knn.hip")]
flags = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fvisibility=hidden"]
subprocess.run(flags + ["-DDSS_NO_XCD_MAP"] + files + ["-o", os.path.join(out, "lib_noxcd.so")], check=True)
which = sys.argv[1]
from dss_amd import _lib
if which == "noxcd":
    _lib.LIB_PATH = os.path.join(out, "lib_noxcd.so")
import bench
dev = torch.device("cuda:0")
for cfg in ("cfg2", "cfg4"):
    if cfg == "cfg2":
        wl = bench.Workload(dev, 1, bench.RowPartition(512, 1, 0))
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests")); import scenes
        pts, nrm, col = scenes.synthetic_cloud(1_000_000, seed=0)
        wl = bench.Workload(dev, 8, bench.RowPartition(1024, 1, 0), cloud=(pts, nrm, col, 1.05e-5))
    r = [wl.fine_kernel_ms(iters=30)[1] for _ in range(3)]
    print(which, cfg, "fine kernel median ms:", [round(x, 4) for x in r])
