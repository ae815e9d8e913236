#!/usr/bin/env python3
"""Developer tool (GPU box): can the WHOLE multi-GPU step -- compute launches AND the three RCCL collectives -- be captured as
one hipGraph?  World size 1 (`nccl` backend on one GPU, like BENCH_FORCE_DIST=1): captures Workload.step(), replays it, compares
image and gradients with the eager step and times both.  -> one JSON line (an error string when the capture is refused)."""
import json, os, socket, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
wl = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0), multi=True)
out = {"rccl": ".".join(map(str, torch.cuda.nccl.version()))}
def quick(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
img, gw, gc = wl.step(); torch.cuda.synchronize()
ref = (img.clone(), gw.clone(), gc.clone())
out["eager_us"] = round(quick(wl.step), 1)
try:
    wl.capture_segments(); out["segments_us"] = round(quick(wl.step_segments), 1)
except Exception as e:  # noqa: BLE001
    out["segments_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
try:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): wl.step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); res = {}
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        res["o"] = wl.step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    o = res["o"]
    out["whole_graph_image_equal"] = bool(torch.equal(o[0], ref[0]))
    out["whole_graph_grad_maxdiff"] = float((o[1] - ref[1]).abs().max())
    out["whole_graph_us"] = round(quick(g.replay), 1)
except Exception as e:  # noqa: BLE001
    out["whole_graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
print(json.dumps(out), flush=True)
dist.destroy_process_group()
