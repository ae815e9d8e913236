#!/usr/bin/env python3
"""Developer tool: launch the fine kernel 10x without and 10x with the fused blend (run under
rocprofv3 --pmc WRITE_SIZE to compare the write traffic of the two epilogues)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dss_amd import _lib, ops
dev = torch.device("cuda:0")
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
S, K = bench.S, bench.K
info = ops.point_setup(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, S, 1.0, 1.0, False, True)
lib = _lib.load()
idx = torch.empty((1, S, S, K), dtype=torch.int32, device=dev); zbuf = torch.empty((1, S, S, K), device=dev)
qv = torch.empty_like(zbuf); occ = torch.empty((1, S, S), device=dev); vis = torch.zeros(wl.P, dtype=torch.uint8, device=dev)
image = torch.empty((1, S, S, 4), device=dev); wsum = torch.empty((1, S, S), device=dev)
nbytes = lib.dss_splat_forward_workspace(1, wl.P, S, K, 1); ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
st = _lib.stream_ptr(dev)
a = (_lib.ptr(info["pts_screen"]), _lib.ptr(info["ellipse_params"]), _lib.ptr(info["cutoff_threshold"]),
     _lib.ptr(info["radii"]), _lib.ptr(wl.first), _lib.ptr(wl.num), 1, wl.P)
_lib.check(lib.dss_splat_bin(a[0], a[3], a[4], a[5], 1, wl.P, S, 0, S, _lib.ptr(ws), nbytes, st), "bin")
novis = "--novis" in sys.argv
for i in range(10):
    _lib.check(lib.dss_splat_fine(*a, bench.THR, S, K, 0, S, _lib.ptr(idx), _lib.ptr(zbuf), _lib.ptr(qv), _lib.ptr(occ),
                                  None if novis else _lib.ptr(vis), _lib.ptr(ws), nbytes, st), "fine")
for i in range(10):
    _lib.check(lib.dss_splat_fine_blend(*a, bench.THR, S, K, 0, S, _lib.ptr(idx), _lib.ptr(zbuf), _lib.ptr(qv),
                                        _lib.ptr(occ), None if novis else _lib.ptr(vis), _lib.ptr(info["scaler"]),
                                        _lib.ptr(wl.colors), 3, _lib.ptr(image), _lib.ptr(wsum), _lib.ptr(ws), nbytes, st), "fineblend")
torch.cuda.synchronize()
