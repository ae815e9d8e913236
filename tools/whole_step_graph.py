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
    torch.cuda.current_stream().wait_stream(side); bench.drain_collective_watchdog()
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
# The layout N > 1 uses by default (tile-row-cyclic bands) reassembles the gathered rows with an index_select on a side stream
# that waits for the asynchronous collective: at world size 1 the bands are uniform and that code does not run -- force it with
# the identity permutation, so that the capture sees the stream fork / join it will see on eight GPUs.
try:
    wl2 = bench.Workload(dev, 1, bench.RowPartition(bench.S, 1, 0), multi=True)
    fx = wl2.fx
    fx.row_index = torch.arange(bench.S, device=dev, dtype=torch.int64)
    fx.full_img = torch.empty((bench.S, 1, bench.S, 4), dtype=torch.float32, device=dev)
    img2 = wl2.step()[0]; torch.cuda.synchronize()
    out["row_index_path_eager_image_equal"] = bool(torch.equal(img2, ref[0]))
    o2 = wl2.capture_whole_step(); wl2.step_whole(); torch.cuda.synchronize()
    out["row_index_path_whole_graph_image_equal"] = bool(torch.equal(o2[0], ref[0]))
    out["row_index_path_whole_graph_us"] = round(quick(wl2.step_whole), 1)
except Exception as e:  # noqa: BLE001
    out["row_index_path_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
print(json.dumps(out), flush=True)
dist.destroy_process_group()
