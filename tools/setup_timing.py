#!/usr/bin/env python3
"""Developer tool (GPU box): phase stamps (100 MHz s_memrealtime) of the first lane of every wavefront of setup_bin_kernel on
the bench scene: cloud lookup | per-point setup | tile rectangle + counter atomics issued | atomics returned | claims + list
stores.  Builds a private -DDSS_FINE_TIMING copy of the library under gpurun_out/."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "gpurun_out"); os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libdss_hip_timing.so"); src = os.path.join(ROOT, "dss_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                "-fno-fast-math", "-fno-slp-vectorize", "-fvisibility=hidden", "-DDSS_FINE_TIMING",
                *sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".hip")), "-o", so], check=True)
from dss_amd import _lib, ops
_lib.LIB_PATH = so
import bench
dev = torch.device("cuda:0"); lib = _lib.load(); lib.dss_debug_set_fine_timing.argtypes = [ctypes.c_void_p]
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
for _ in range(5): wl.step()
torch.cuda.synchronize()
buf = torch.zeros((16384, 12), dtype=torch.int64, device=dev)
assert lib.dss_debug_set_fine_timing(ctypes.c_void_p(buf.data_ptr())) == 0
for rep in range(3):
    buf.zero_(); torch.cuda.synchronize()
    wl.step(); torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.float64)[8192:]
    used = t[:, 0] > 0
    t = t[used]
    t0 = t[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    names = {0: "start", 1: "cloud found", 7: "inputs arrived", 6: "arithmetic done", 2: "setup done (stores issued)",
             8: "tile rectangle known", 9: "addresses ready", 3: "counter atomics issued", 4: "atomics returned",
             5: "claims + list stores done"}
    print("rep %d: %d wavefronts" % (rep, len(t)))
    for i, nm in names.items():
        col = t[:, i][t[:, i] > 0]
        if len(col):
            print("  %-30s mean %6.2f  p90 %6.2f  max %6.2f us (%d)" % (nm, us(col).mean(), np.percentile(us(col), 90), us(col).max(), len(col)))
