#!/usr/bin/env python3
"""Developer tool (GPU box): A/B of compile-time variants.  Builds a private copy of the library under gpurun_out/ for
every set of -D flags given and times the fine pass (HIP events, bench.Workload.fine_kernel_ms) and the whole bench step.
    python tools/ab_macro.py "" "-DDSS_EXP_PRIO=1" ["-DA -DB" ...] [--cfg cfg2|cfg3|cfg4|cfg5]
The shipped libdss_hip.so is never touched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dss_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "ab")

CHILD = r'''
import sys, os, json, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
from dss_amd import _lib
_lib.LIB_PATH = %(so)r
import bench, scenes
cfg = %(cfg)r
dev = torch.device("cuda:0")
if cfg == "cfg2":
    wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0)); S = bench.S
else:
    P, S, N = {"cfg4": (1000000, 1024, 8), "cfg5": (4000000, 2048, 1), "cfg3": (99790, 512, 8)}[cfg]
    pts, nrm, col = scenes.synthetic_cloud(P, seed=0)
    h = scenes.global_h(pts[:: max(1, P // 200000)]) * (200000 / P if P > 200000 else 1.0)
    wl = bench.Workload(dev, N, bench.RowPartition(S, 1, 0), cloud=(pts, nrm, col, float(np.clip(h, 5e-6, 1e-3))))
for _ in range(5): wl.step()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(50 if cfg in ("cfg2", "cfg3") else 10): wl.step()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / (50 if cfg in ("cfg2", "cfg3") else 10) * 1e3)
gbest = None
if cfg == "cfg2":
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): wl.step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        wl.step()
    gbest = 1e9
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): g.replay()
        torch.cuda.synchronize()
        gbest = min(gbest, (time.perf_counter() - t0) / 200 * 1e3)
fm = [wl.fine_kernel_ms(iters=30) for _ in range(3)]
gm = min(wl.backward_gather_ms(iters=30)[0] for _ in range(3))
print(json.dumps({"gather_ms": round(gm, 5), "step_ms_graph_best": None if gbest is None else round(gbest, 5), "step_ms_eager_best": round(best, 5), "fine_ms_mean": round(min(f[0] for f in fm), 5),
                  "fine_ms_median": round(min(f[1] for f in fm), 5)}))
'''


def main():
    args = [a for a in sys.argv[1:]]
    cfg = "cfg2"
    if "--cfg" in args:
        i = args.index("--cfg")
        cfg = args[i + 1]
        del args[i:i + 2]
    os.makedirs(OUT, exist_ok=True)
    for k, flags in enumerate(args or [""]):
        so = os.path.join(OUT, "libdss_hip_ab%d.so" % k)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                        "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", *flags.split(),
                        *sorted(os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".hip")), "-o", so], check=True)
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "so": so, "cfg": cfg}], capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
        print("%-40s %s" % (flags or "(baseline)", line), flush=True)
        os.remove(so)


if __name__ == "__main__":
    main()
