#!/usr/bin/env python3
"""Developer tool (GPU box): per-rank COMPUTE time of the multi-GPU bench step, emulated on one GPU: G cameras, each
rank's rows (contiguous equal bands and the tile-row-cyclic partition) through the SAME object `bench.py --gpus G` drives
(dss_amd.sharded.RowShardedRender): forward of the rows, the image loss of the rows, backward, clip + projection.  What the
collectives would deliver (visibility union, all-reduced loss partials, all-gathered alpha-gradient plane) is computed once,
untimed, from the full image.  No collectives: this is the part of the G-GPU step that RCCL time is added to.  Per rank the step is timed twice: eager launches (host-bound at these sizes) and as a hipGraph replay (device time).

    python tools/band_timing.py [G] [cfg2|cfg4|cfg5] -> one JSON line: per-rank microseconds, max / min spread, for both layouts
(cfg2: G cameras x 32,684 points @512^2, the weak-scaling workload of `bench.py --gpus G`; cfg4 / cfg5: BASELINE configs[3] /
configs[4] -- 8 x 1M points @1024^2 / 4M points @2048^2 -- with their rows shared by G ranks, `bench.py --gpus G --workload`)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dss_amd import ops  # noqa: E402
from dss_amd.distributed import RowPartition, balanced_bounds, fitted_bounds, rebalanced_bounds  # noqa: E402
from dss_amd.sharded import RowShardedRender  # noqa: E402

from dss_amd import _lib  # noqa: E402
if os.environ.get("BAND_TPW"):       # development A/B: DSS_OPT_BACKWARD_TPW
    _lib.set_option(_lib.OPT_BACKWARD_TPW, int(os.environ["BAND_TPW"]))
if os.environ.get("BAND_FUSED"):     # development A/B: DSS_OPT_BACKWARD_FUSED (1 = the round-3 launch sequence on the band)
    _lib.set_option(_lib.OPT_BACKWARD_FUSED, int(os.environ["BAND_FUSED"]))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
which = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
dev = torch.device("cuda:0")
S, K = bench.S, bench.K
if which == "cfg2":
    wl = bench.Workload(dev, G, RowPartition(S, 1, 0))   # G cameras, single-rank object (no process group needed)
else:
    cloud, S, cams = bench.large_cloud(which)
    wl = bench.Workload(dev, cams, RowPartition(S, 1, 0), cloud=cloud)


# clouds above 2M splats: the renderer-owned cached point order, like `bench.py --workload cfg4|cfg5` (every k-th call sorts
# and saves, the others bin through the saved order)
ORDER_REFRESH = int(os.environ.get("BAND_ORDER_REFRESH", "16")) if wl.P > 2_000_000 else 0


def fwd(rows):
    return ops.render_forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors, S, K,
                              bench.CUTOFF, bench.THR, bench.SIGMA, False, True, rows=rows, order_refresh=ORDER_REFRESH,
                              band_outputs_only=rows is not None and os.environ.get("BAND_FULL_OUTPUTS") != "1")


def quick(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


out = {"G": G, "workload": which, "cameras": wl.N, "points_per_cloud": wl.Pc, "image_size": S}
GRADIENT = os.environ.get("BAND_GRADIENT", "auto")   # bench.py's default: chosen from the bytes on the critical path
if GRADIENT == "auto":
    from dss_amd.sharded import choose_gradient_exchange
    GRADIENT = choose_gradient_exchange(wl.N, wl.Pc, wl.P, S, 3, max(G, 2), os.environ.get("BAND_LOSS", "band") != "replicated", True)
out["gradient"] = GRADIENT
N_IT = 60 if which == "cfg2" else 16
REPLICATED = os.environ.get("BAND_LOSS", "band") == "replicated"   # loss on the gathered image on every rank (unmodified loops)
out["loss"] = "replicated" if REPLICATED else "band"
OWNER = GRADIENT == "owner"   # gradient exchange of the step (bench.py BENCH_GRADIENT): owner | bucket
TRACE = os.environ.get("BAND_TRACE") == "1"   # under rocprofv3 --kernel-trace: the cyclic partition, rank 3, eager launches only
LAYOUTS = os.environ.get("BAND_LAYOUTS", "bands,balanced,cyclic").split(",")
TRACE_RANK = int(os.environ.get("BAND_TRACE_RANK", "3"))
for layout in ((os.environ.get("BAND_TRACE_LAYOUT", "cyclic"),) if TRACE else LAYOUTS):
    bounds = None
    if layout == "balanced":
        # contiguous bands with (about) equal load: row weight = occupied pixels per image row, summed over the cameras, of a
        # full render (every rank holds the gathered image of the previous step: the same weights everywhere, no collective)
        w = fwd(None)["occupancy"].sum(dim=(0, 2)).double().cpu() + float(os.environ.get("BAND_ROW_BIAS", "0"))
        bounds = balanced_bounds(w, G, align=8, min_rows=8)
        out["balanced_bounds"] = bounds
    if layout == "fitted":
        # contiguous bands from the MEASURED times of the two layouts above (dss_amd.distributed.fitted_bounds)
        w = fwd(None)["occupancy"].sum(dim=(0, 2)).double().cpu()
        equal = [RowPartition(S, G, r).rows[0] for r in range(G)] + [S]
        bounds, fit = fitted_bounds(w, [(equal, out["bands"]["graph_us"]), (out["balanced_bounds"], out["balanced"]["graph_us"])], G)
        out["fitted_bounds"], out["fitted_model_F_a_b"] = bounds, [float("%.4g" % v) for v in fit]
    if layout.startswith("rebalanced"):
        # measured-time rebalancing (dss_amd.distributed.rebalanced_bounds), starting from the fitted layout; rebalanced2 = a second step
        prev = "fitted" if layout == "rebalanced" else "rebalanced"
        bounds = rebalanced_bounds(out[prev + "_bounds"], out[prev]["graph_us"], out["fitted_model_F_a_b"][0])
        out[layout + "_bounds"] = bounds
    parts = [RowPartition(S, G, r, cyclic=(layout == "cyclic"), bounds=bounds) for r in range(G)]
    # (union of the ranks' visibility flags = the flags of the full render: one call instead of G band renders)
    full = fwd(None)
    vis_all = full["visible"].clone()
    # the CAUSAL step of bench.py --gpus G (dss_amd.sharded.RowShardedRender): the image gradient is the reference's image loss
    # of the rank's own rows; what the collectives would deliver is computed once, untimed, from the full image -- the
    # all-reduced loss partials and (owner form) the all-gathered alpha-gradient plane
    img0 = torch.roll(full["image"], shifts=(5, 9), dims=(1, 2))
    t_rgb, t_mask = img0[..., :3].contiguous(), (img0[..., 3] > 0).float().contiguous()
    red = ops.image_loss_band_partials(full["image"].contiguous(), t_rgb, t_mask, (0, S))
    g_full, _ = ops.image_loss_band_backward_partials(full["image"].contiguous(), t_rgb, t_mask, (0, S), 1.0, 1.0, red)
    alpha_full = g_full[..., 3].contiguous()          # (N, S, S)
    full_image = full["image"].contiguous()
    eager, graph = [], []
    for p in (parts[TRACE_RANK:TRACE_RANK + 1] if TRACE else parts):
        eng = RowShardedRender(p, wl.N, wl.Pc, wl.P, S, K, 3, dev, True, bench.CUTOFF, bench.SIGMA, bench.THR,
                               gradient=GRADIENT, features_shared=True, force=False)
        eng.active = False                            # no process group here: the collectives are emulated by their results
        bt = tuple(x.contiguous() for x in ops.band_targets(t_rgb, t_mask, p.rows))
        alpha_needed = OWNER and G > 1
        if alpha_needed:
            from dss_amd.distributed import AlphaPlaneExchange
            eng.alpha_x = AlphaPlaneExchange(p, wl.N, dev)
            pos = torch.tensor(p.gather_index(), dtype=torch.int64, device=dev)
            eng.alpha_x.recv.zero_()
            eng.alpha_x.recv[pos] = alpha_full.permute(1, 0, 2)     # what the all-gather would leave behind

        def step_replicated():
            # BAND_LOSS=replicated: what an UNMODIFIED training loop does with `row_output="full"` -- every rank evaluates the
            # loss on the whole gathered image (here: the full render, as the all-gather would deliver it) and hands the full
            # gradient to the backward (owner form: no further exchange; bucket form: its own rows of it)
            f = eng.forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors,
                            order_refresh=ORDER_REFRESH)
            losses, sums = ops.image_loss_forward(full_image, t_rgb, t_mask, 1.0, 1.0)
            g = ops.image_loss_backward(full_image, t_rgb, t_mask, 1.0, 1.0, sums)
            eng.bwd_begin(g, full=True)
            eng.bwd_compute(bench.RADII_S, bench.CLIP, wl.world, wl.M, wl.V, wl.first, wl.num, f=f, vis_all=vis_all)
            return eng.bwd_finish(bench.CLIP, wl.world, wl.M, wl.V, wl.first, wl.num)

        def step():
            if REPLICATED:
                return step_replicated()
            f = eng.forward(wl.world, wl.normals, wl.h, wl.M, wl.V, wl.znear, wl.zfar, wl.first, wl.num, wl.colors,
                            order_refresh=ORDER_REFRESH)
            band = eng.band_image                                                                # (strided view of the send buffer)
            ops.image_loss_band_partials(band, t_rgb, t_mask, p.rows, band_targets=bt)           # (its all-reduced form: `red`)
            g_band, _ = ops.image_loss_band_backward_partials(band, t_rgb, t_mask, p.rows, 1.0, 1.0, red, band_targets=bt,
                                                              alpha_out=eng.alpha_send_view())   # (owner: + the alpha channel, packed)
            eng.bwd_begin(g_band, alpha_packed=True, full=False)
            eng.bwd_compute(bench.RADII_S, bench.CLIP, wl.world, wl.M, wl.V, wl.first, wl.num, f=f, vis_all=vis_all)
            return eng.bwd_finish(bench.CLIP, wl.world, wl.M, wl.V, wl.first, wl.num)
        eager.append(quick(step, N_IT))
        if TRACE:
            graph.append(eager[-1])
            continue
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg, stream=side):
            step()
        graph.append(quick(cg.replay, N_IT))
    out[layout] = {"eager_us": [round(x, 1) for x in eager], "graph_us": [round(x, 1) for x in graph],
                   "graph_max_over_min": round(max(graph) / min(graph), 3), "graph_max_us": round(max(graph), 1),
                   "eager_max_us": round(max(eager), 1)}
if not TRACE:   # (a kernel trace holds one rank's band step only)
    one = RowPartition(S, 1, 0)
    wl1 = bench.Workload(dev, 1, one) if which == "cfg2" else wl
    out["single_gpu_step_us"] = {"eager": round(quick(wl1.step, N_IT), 1)}
print(json.dumps(out))
