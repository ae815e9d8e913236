import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from dss_amd import ops
dev = torch.device("cuda:0")
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0))
one = torch.zeros(1, dtype=torch.int64, device=dev); cnt = torch.full((1,), wl.Pc, dtype=torch.int64, device=dev)
h_buf = wl.h
def knn_only():
    return ops.cloud_mean_clamp(ops.knn_kth_sqdist(wl.world, one, cnt, 7), one, cnt, 0.5, 5e-5, 1e-3, 0.5e-3, 7)
def both():
    h_buf.copy_(knn_only().expand_as(h_buf)); return wl.step()
def quick(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
def graph(fn, unroll=1):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(unroll): fn()
    return g
print("eager knn", quick(knn_only), "eager step", quick(wl.step), "eager both", quick(both))
for u in (1, 10):
    print("unroll", u, "graph knn", quick(graph(knn_only, u).replay, 20) / u, "graph step", quick(graph(wl.step, u).replay, 20) / u, "graph both", quick(graph(both, u).replay, 20) / u)
