#!/usr/bin/env python3
"""bench.py -- throughput of the EWA splatting hot path on MI355X (contract: see README / task).

One "step" = one forward + backward pass of the whole hot path over one batch of synthetic input
already resident in HBM:
    point_setup -> splat_forward (bin + fine) -> blend_forward -> blend_backward -> backward_radius -> occ_backward -> clip
    -> project_backward
Workload at N=1 = BASELINE.json configs[1]: "bunny ~30k" (bunny-8000 x4 tangent-plane jitter =
32,684 points), 1 camera, 512x512, K=5, fwd+bwd with grad_out = randn(seed 1) on RGBA.
For N GPUs the batch holds N cameras (ring, azim = 45 deg * k) and every image is row-partitioned
across the N ranks (weak scaling: rows x cameras per rank is constant; tile-row-cyclic: rank g renders
the 8-row tile rows g, g + N, ...).  The N > 1 step is the object `SurfaceSplattingRenderer(row_partition=...)` drives
(dss_amd.sharded.RowShardedRender) and it is CAUSAL -- the image gradient comes from data the rank holds at that moment:
    forward of the rank's rows -> [visibility all-reduce | image all-gather, asynchronous] -> the reference's image loss
    (Trainer.calc_dr_loss) of the rank's own rows, per-image sums all-reduced -> [owner form: all-gather of the alpha-gradient
    plane] -> backward of the rank's rows -> clip + projection -> [all-reduce of the world-space gradient sums]
    (BENCH_GRADIENT=bucket: backward -> [all-reduce of every (camera, point) pair's partial sums] -> clip + projection)

Metric: Msplats/s = (cameras * points per cloud) / step time, whole job.

How the timed step is launched (recorded in `config.launch`): one GPU -- hipGraph replays of the identical launch sequence
(`graph_x10`: ten steps per graph launch) unless --mode says otherwise; --workload cfg4|cfg5 -- two graphs, the step that
sorts and saves the point order and the step that reuses it, replayed in the renderer's rhythm (`graph_save_reuse`); N > 1
(or BENCH_FORCE_DIST=1) over RCCL -- the WHOLE step, launches and collectives, as one graph (`graph_step`) after it
has been checked against the eager step on every rank, else graphs of the compute segments between host-issued collectives.
Environment switches (development A/B; none is needed for the contract): BENCH_FORCE_DIST=1 (multi-GPU path at world size
1), BENCH_DIST_BACKEND=gloo (CPU collectives, tests), BENCH_NO_WHOLE_GRAPH=1 (segments instead of the whole-step graph),
BENCH_IMAGE_LATE=1 (image collective issued behind the backward), BENCH_EXCHANGE=overlap|fold|auto (end-of-forward exchange),
BENCH_GRADIENT=auto|owner|bucket (gradient exchange), BENCH_ROW_PARTITION=cyclic|bands (row layout), BENCH_ORDER_REFRESH=k (period of the cached point order of
the large workloads, 0 = sort in every step; default 16), BENCH_BACKWARD_FUSED / BENCH_BACKWARD_TPW (DSS_OPT_* of the
library), DSS_AMD_CALLING_THREAD_BACKWARD=1 (backward on the calling thread for the API figures; the scoped form is
`with dss_amd.calling_thread_backward():`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dss_amd import _lib, ops  # noqa: E402
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform  # noqa: E402
from dss_amd.distributed import RowPartition, gather_rows  # noqa: E402
from dss_amd.sharded import RowShardedRender  # noqa: E402

S, K, THR, RADII_S, CLIP, CUTOFF, SIGMA = 512, 5, 0.05, 5.0, 0.05, 1.0, 1.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def bunny_cloud():
    """bunny-8000 (tests/golden/clouds.npz, converted from the reference fixture) x4 jitter upsample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, 4, seed=0)
    col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
    return pts, nrm, col, None  # h (variance scale) is computed on the GPU by the HIP kNN in Workload


# --workload: the other two BASELINE configs that name multi-GPU row sharding (points per cloud, image side, cameras);
# the camera count is FIXED there (strong scaling: N ranks share the rows of the same job), unlike the headline workload
LARGE_WORKLOADS = {"cfg4": (1_000_000, 1024, 8), "cfg5": (4_000_000, 2048, 1), "cfg3": (99_790, 512, 8)}


def large_cloud(which, morton=False):
    """BASELINE configs[3] / configs[4] (and the size of configs[2]): the synthetic generator of SURVEY 8(d)
    (tests/scenes.py::synthetic_cloud), randomly ordered, with a density-scaled variance scale h (kNN-7 statistic of a
    200k-point subsample scaled by the density ratio: the reference's clamp [5e-5, 1e-3] would turn a 4M-point cloud
    into 20-pixel splats).  -> (points, normals, colours, h), S, cameras"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    P, S_, N = LARGE_WORKLOADS[which]
    pts, nrm, col = scenes.synthetic_cloud(P, seed=0)
    h = scenes.large_cloud_h(pts)   # (shared with tests/test_gpu_named_configs.py: the benched scene is the parity-tested one)
    if morton:
        from dss_amd.cloud import spatial_order
        order = spatial_order(torch.from_numpy(pts)).numpy()
        pts, nrm, col = pts[order].copy(), nrm[order].copy(), col[order].copy()
    return (pts, nrm, col, h), S_, N


def drain_collective_watchdog():
    """Before a graph capture in a process that has issued RCCL collectives: wait until the process group's watchdog thread
    has dropped the finished ones.  It polls the end events of the outstanding collectives every 100 ms; those events were
    recorded on RCCL's internal stream, and if a poll falls into the capture of a step whose collectives put that stream into
    capture mode, HIP refuses the query (hipErrorCapturedEvent, "an event last recorded in a capturing stream") and the
    watchdog takes the process down -- seen once in ~40 runs of `BENCH_FORCE_DIST=1 BENCH_EXCHANGE=auto bench.py --gpus 1`.
    After a device synchronisation every collective has finished; three poll intervals later none is on the list."""
    torch.cuda.synchronize()
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        time.sleep(0.35)


class Workload:
    def __init__(self, device, n_cams, part: RowPartition, cloud=None, multi=None, fold=False):
        pts, nrm, col, h = bunny_cloud() if cloud is None else cloud
        self.dev, self.N, self.part = device, n_cams, part
        # multi: take the multi-GPU path (dss_amd.sharded.RowShardedRender: send buffers, the exchanges, the band loss, the
        # gradient reduction); forced at world size 1 by BENCH_FORCE_DIST=1 so that a one-GPU box executes the RCCL code path.
        # multi="local": the same causal step (render -> image loss -> backward) on ONE rank without any collective -- the
        # single-rank reference of the tests and the like-for-like single-GPU figure `value_with_image_loss`
        self.local = multi == "local"
        if self.local and part.world_size != 1:
            raise ValueError("multi='local' is the one-rank form of the step")
        self.multi = part.world_size > 1 if multi is None else bool(multi)
        # renderer-owned cached point order (clouds above 2M points: cfg4 / cfg5): every k-th step sorts and saves the
        # order, the others bin through it (`SurfaceSplattingRenderer(order_refresh=k)`); 0 = every step sorts
        # (below 2M splats the library bins directly and ignores the order flags; BENCH_ORDER_REFRESH set explicitly is passed on
        # all the same: A/B builds with another threshold, tools/ab_bench.py)
        self.order_refresh = int(os.environ.get("BENCH_ORDER_REFRESH", "16")) if (
            n_cams * pts.shape[0] > 2_000_000 or "BENCH_ORDER_REFRESH" in os.environ) else 0
        self.force_order = None   # "save" / "reuse": this step's kind is fixed (the two graphs of the cached-order mode)
        S = part.S  # image side (module constant S for the benchmark; tools/bench_large.py passes others)
        self.Pc = pts.shape[0]
        self.P = self.N * self.Pc
        t = lambda a: torch.from_numpy(a).to(device)
        self.world, self.normals = t(pts), t(nrm)
        self.colors = t(col).repeat(self.N, 1).contiguous()  # packed (N*Pc,3) features of the extended cloud
        if h is None:
            # Vrk_invariant scale (rasterizer.py:310-326) from the HIP grid kNN; an input of the step, computed
            # once outside the timed region (the reference caches it the same way with refresh=False)
            one = torch.zeros(1, dtype=torch.int64, device=device)
            cnt = torch.full((1,), self.Pc, dtype=torch.int64, device=device)
            h = float(ops.cloud_mean_clamp(ops.knn_kth_sqdist(self.world, one, cnt, 7), one, cnt, 0.5, 5e-5, 1e-3,
                                           0.5e-3, 7).item())
        self.h = torch.full((self.N,), h, device=device)
        R, T = look_at_view_transform(2.0, 30.0, [45.0 + 45.0 * k for k in range(self.N)])
        cam = FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R, T=T)
        self.M = cam.get_full_projection_transform().get_matrix().to(device).contiguous()
        self.V = cam.get_world_to_view_transform().get_matrix().to(device).contiguous()
        self.znear = torch.full((self.N,), 0.1, device=device)
        self.zfar = torch.full((self.N,), 100.0, device=device)
        self.first = torch.arange(self.N, device=device, dtype=torch.int64) * self.Pc
        self.num = torch.full((self.N,), self.Pc, device=device, dtype=torch.int64)
        self.S = S
        self.grad_out = None
        if not self.multi:
            # one GPU (the metric's configuration): d loss / d RGBA = randn(seed 1), an input of the step
            g = torch.Generator(device="cpu").manual_seed(1)
            self.grad_out = torch.randn((self.N, S, S, 4), generator=g).to(device)
        if not self.multi and self.N == 1 and part.world_size == 1:
            self._plan = ops.FusedPlan(device, 1, self.Pc, self.P, S, K, 3, True, False, False, False, CUTOFF, SIGMA, THR,
                                       want_zbuf=True)
            self._plan.order_refresh = self.order_refresh
        if self.multi:
            # multi-GPU: ONE object holds the rank's share of the step -- dss_amd.sharded.RowShardedRender, the same one
            # `SurfaceSplattingRenderer(row_partition=...)` drives.  The forward kernel writes its RGBA band and visibility flags
            # straight into the exchange's send buffers.  The step is CAUSAL: the image gradient is the reference's image loss
            # (Trainer.calc_dr_loss, trainer.py:332-376: masked L1 on RGB + L1 + 0.01 IoU on the occupancy) of the rank's OWN
            # band against fixed targets, its per-image sums all-reduced (dss_amd.distributed.band_image_loss's kernels).
            # Gradient exchange (BENCH_GRADIENT=auto|owner|bucket; auto = dss_amd.sharded.choose_gradient_exchange, from the
            # bytes each form puts on the critical path: bucket at the metric's configuration, owner at configs[3] / [4]): "owner" -- the rank whose band holds a point's centre row computes
            # the pair's WHOLE position gradient; it needs the occupancy gradient of all rows, so the ranks all-gather that one
            # channel (N S^2 4 bytes) between the loss and the backward; clip + projection then run before ONE all-reduce of
            # the world-space sums (Pc x 6 floats).  "bucket" -- partial sums of every (camera, point) pair from the band's own
            # gradient, ONE all-reduce of N Pc x 6 floats, clip + projection behind it.
            # End-of-forward exchange: "overlap" -- visibility all-reduce (critical) + asynchronous image all-gather on a second
            # communicator (nobody in the step reads the full image: it is the step's output) -- or "fold", see main().
            self.engine = RowShardedRender(part, self.N, self.Pc, self.P, S, K, 3, device, True, CUTOFF, SIGMA, THR,
                                           gradient=os.environ.get("BENCH_GRADIENT", "auto"), features_shared=True,
                                           fold=fold, force=not self.local,
                                           late_image=os.environ.get("BENCH_IMAGE_LATE", "0") == "1")
            self.owner = self.engine.owner
            # targets of the loss: the scene's own render shifted by a few pixels (identical on every rank, outside the timed
            # region): the silhouette term then pulls on every point near the outline, like an early training iteration
            f0 = ops.render_forward(self.world, self.normals, self.h, self.M, self.V, self.znear, self.zfar, self.first,
                                    self.num, self.colors, S, K, CUTOFF, THR, SIGMA, False, True, want_zbuf=False)
            img0 = torch.roll(f0["image"], shifts=(5, 9), dims=(1, 2))
            self.target_rgb = img0[..., :3].contiguous()
            self.target_mask = (img0[..., 3] > 0).float().contiguous()
            self.band_targets = tuple(x.contiguous() for x in ops.band_targets(self.target_rgb, self.target_mask, part.rows))
            del f0, img0
            self._st = {}

    # the exchange object of the current form (tests and the diagnostics look at it)
    fx = property(lambda self: self.engine.fx)

    def set_exchange(self, fold: bool):
        """select the exchange form of the multi-GPU step (captured graphs hold the buffers of the form they were captured with)"""
        self.engine.set_exchange(fold)
        self._graphs = self._whole = None

    # ---- the stages of the multi-GPU step: compute ("c") and collectives ("x") alternate ----------------------------------
    def _st_forward(self):
        kw = {"order_refresh": self.order_refresh} if self.force_order is None else {"workspace_state": self._order_ws_state()}
        self._st["f"] = self.engine.forward(self.world, self.normals, self.h, self.M, self.V, self.znear, self.zfar, self.first,
                                            self.num, self.colors, **kw)

    def _st_exchange(self):
        e = self.engine
        self._st["vis"] = e.start_exchange()   # (a static buffer either way: the flags in place, or the folded form's union)

    def _st_loss_sums(self):
        # (the band as the forward wrote it: the strided (N, rows, S, 4) view of the exchange's send buffer, no copy)
        self._st["band"] = band = self.engine.band_image
        self._st["sums"] = ops.image_loss_band_partials(band, self.target_rgb, self.target_mask, self.part.rows,
                                                        band_targets=self.band_targets)

    def _st_loss_reduce(self):
        if self.engine.active:
            dist.all_reduce(self._st["sums"], op=dist.ReduceOp.SUM)      # block partials of the five sums: 2560 N bytes

    def _st_loss_grad(self):
        st = self._st
        # owner form on a band: the loss kernel also writes the alpha channel of its gradient into the send buffer of the
        # alpha-plane exchange
        st["g_band"], st["losses"] = ops.image_loss_band_backward_partials(
            st["band"], self.target_rgb, self.target_mask, self.part.rows, 1.0, 1.0, st["sums"], band_targets=self.band_targets,
            alpha_out=self.engine.alpha_send_view())
        self.engine.bwd_begin(st["g_band"], alpha_packed=True, full=False)

    def _st_backward(self):
        self.engine.bwd_compute(RADII_S, CLIP, self.world, self.M, self.V, self.first, self.num, f=self._st["f"],
                                vis_all=self._st["vis"])
        if self.engine.late_image:
            self.engine.start_image()

    def _st_finish(self):
        self._st["out"] = self.engine.bwd_finish(CLIP, self.world, self.M, self.V, self.first, self.num)

    def _st_image(self):
        self._st["image"] = self.engine.full_image()

    def stages(self):
        """[(kind, label, callable)] of one multi-GPU step, in order; kind "c" = kernels of this rank, "x" = a collective"""
        e = self.engine
        alpha = e.owner and self.part.world_size > 1
        return ([("c", "forward", self._st_forward), ("x", "visibility_allreduce", self._st_exchange),
                 ("c", "loss_sums", self._st_loss_sums), ("x", "loss_allreduce", self._st_loss_reduce),
                 ("c", "loss_gradient", self._st_loss_grad)]
                + ([("x", "alpha_allgather", e.bwd_exchange_alpha)] if alpha else [])
                + [("c", "backward", self._st_backward), ("x", "gradient_allreduce", e.bwd_reduce),
                   ("c", "projection", self._st_finish), ("x", "image_allgather", self._st_image)])

    def step(self, ev=None):
        """one forward + backward.  `ev` (multi-GPU diagnostics): a list that receives (label, event) marks recorded on the
        current stream between the compute stages and the collectives."""
        def mark(label):
            if ev is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((label, e))
        p = self.part
        S = self.S
        mark("start")
        if not self.multi and self.N == 1 and p.world_size == 1:
            # one GPU, one camera (the metric's configuration): the two fused entry points through ops.FusedPlan -- the same
            # C calls as ops.render_forward / ops.render_backward below with the host work of a call cut down (one arena for the
            # 13 outputs, prebuilt argument lists): what `SurfaceSplattingRenderer` itself uses
            plan = self._plan
            plan.force_state = self._order_ws_state()
            arena = plan.forward(self.world, self.normals, self.h, self.M, self.V, self.znear, self.zfar, self.first, self.num,
                                 self.colors)
            g_feat, g_world = plan.backward(arena, self.grad_out, self.first, self.num, RADII_S, CLIP, self.world, self.M)
            mark("projection_compute")
            return plan.image(arena), g_world, g_feat
        if not self.multi:
            # one GPU, several cameras (emulation tools): fused forward, fused backward, projection + colour reduction
            f = ops.render_forward(self.world, self.normals, self.h, self.M, self.V, self.znear, self.zfar, self.first,
                                   self.num, self.colors, S, K, CUTOFF, THR, SIGMA, False, True, rows=p.rows,
                                   **({"order_refresh": self.order_refresh} if self.force_order is None else
                                      {"workspace_state": self._order_ws_state()}))
            image = gather_rows(f["image"], p)
            g_feat, g_pts = ops.render_backward(self.grad_out, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"],
                                                f["radii"], f["visible"], self.first, self.num, RADII_S, CLIP)
            g_world, g_col = ops.project_backward(self.world, self.M, self.V, self.first, self.num, g_pts, f["valid"], True,
                                                  clip=-1.0, grad_features=g_feat)
            mark("projection_compute")
            return image, g_world, g_col
        for kind, label, fn in self.stages():
            fn()
            mark(("wait_" + label) if kind == "x" else (label + "_compute"))
        return self._st["image"], self._st["out"][0], self._st["out"][1]

    def _order_ws_state(self):
        """workspace_state of the forward when the kind of the step is fixed (None: the renderer's own bookkeeping decides)"""
        from dss_amd import _lib
        return None if self.force_order is None else (1 | (_lib.WS_ORDER_SAVE if self.force_order == "save" else _lib.WS_ORDER_REUSE))

    # ---- multi-GPU: the compute between the collectives as hipGraphs ---------------------------------
    def capture_segments(self):
        """One graph per run of consecutive compute stages of `stages()`, replayed around the host-issued collectives: the
        multi-rank step then costs the host a handful of graph launches + the collective calls instead of ~15 kernel
        launches through Python."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self.step()
        torch.cuda.current_stream().wait_stream(side)
        drain_collective_watchdog()
        plan, run = [], []
        for kind, label, fn in self.stages():
            if kind == "c":
                run.append(fn)
                continue
            if run:
                plan.append(("c", run))
                run = []
            plan.append(("x", fn))
        if run:
            plan.append(("c", run))
        # thread-local capture mode: the process group's watchdog thread may query events while this thread captures; in
        # the default (global) mode such a call from ANOTHER thread invalidates the capture.  The collectives between the
        # segments are issued eagerly while capturing, so that every segment is captured on the state its predecessors left.
        graphs = []
        with torch.cuda.stream(side):
            for kind, what in plan:
                if kind == "x":
                    what()
                    graphs.append(("x", what))
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    for fn in what:
                        fn()
                g.replay()
                graphs.append(("c", g))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graphs = graphs

    def step_segments(self, ev=None):
        """`step` for N > 1 with the compute segments replayed as graphs (same launches, same collectives)"""
        def mark(label):
            if ev is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((label, e))
        mark("start")
        i = 0
        for kind, what in self._graphs:
            if kind == "c":
                what.replay()
                mark("segment%d_compute" % i)
            else:
                what()
                mark("wait_collective%d" % i)
            i += 1
        return self._st["image"], self._st["out"][0], self._st["out"][1]

    # ---- multi-GPU: the WHOLE step as one hipGraph -- launches AND the RCCL collectives -------------------------------------
    def capture_whole_step(self, unroll=1):
        """One graph for the step of `step` (multi branch): RCCL collectives are stream operations like any other and can be
        captured with the kernels around them (measured at world size 1, tools/whole_step_graph.py: 80 us per step against
        120 with the three graph segments and 152 eagerly -- the host then issues ONE graph launch instead of three launches +
        three collective calls).  -> the outputs of the captured step (static tensors the replays overwrite)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self.step()
        torch.cuda.current_stream().wait_stream(side)
        drain_collective_watchdog()
        g = torch.cuda.CUDAGraph()
        res = {}
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(unroll):   # (`unroll` consecutive steps per graph launch, like graph_x10 on one GPU)
                res["out"] = self.step()
        torch.cuda.synchronize()
        self._whole = (g, res["out"])
        return res["out"]

    def step_whole(self, ev=None):
        """`step` for N > 1 as ONE graph replay (see capture_whole_step)"""
        self._whole[0].replay()
        return self._whole[1]

    def dist_timing(self, iters=20):
        """Where a multi-GPU step spends its time, per rank: event-timed stages of `iters` eager steps (microseconds,
        means).  *_compute = kernels of this rank's band; wait_* = time the stream spends in (waiting for) each
        collective.  -> dict of label -> us, plus 'compute_us'."""
        for _ in range(3):
            self.step()
        acc = {}
        for _ in range(iters):
            ev = []
            self.step(ev)
            torch.cuda.synchronize()
            for (l0, e0), (l1, e1) in zip(ev[:-1], ev[1:]):
                acc[l1] = acc.get(l1, 0.0) + e0.elapsed_time(e1) * 1e3 / iters
        acc["compute_us"] = sum(v for k, v in acc.items() if k.endswith("_compute"))
        return acc

    # ---- per-kernel timing with HIP events on the launch stream (torch's current stream) ------
    @staticmethod
    def _event_ms(run, iters=50, warm=5):
        for _ in range(warm):
            run()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for s, e in ev:
            s.record()
            run()
            e.record()
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in ev)
        return float(np.mean(ms)), float(ms[len(ms) // 2])

    @staticmethod
    def _batch_ms(run, batch=20, reps=10):
        """Mean duration of one launch of `run` from `batch` back-to-back launches between ONE pair of events: the ~3-4 us an
        event pair adds to a single 17 us kernel is spread over `batch` launches (the dispatch gaps between the launches
        stay in: still an upper bound of the kernel's own duration).  Only meaningful when the host issues `run` faster
        than the GPU executes it (a prebuilt ctypes call: ~2 us)."""
        for _ in range(3):
            run()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for s, e in ev:
            s.record()
            for _ in range(batch):
                run()
            e.record()
        torch.cuda.synchronize()
        return float(np.mean([s.elapsed_time(e) / batch for s, e in ev]))

    def fine_kernel_ms(self, iters=50):
        """The forward's dominant kernel alone, exactly as the step launches it: dss_render_forward with DSS_WS_BINNED
        repeats only its second launch (fine pass + fused blend on the packed splat records) on the tile lists a
        DSS_WS_UNKNOWN call with the same inputs left in the workspace -- one kernel per timed call, issued through a
        prebuilt ctypes call so that no host work sits between the two events."""
        lib = _lib.load()
        p, S, dev = self.part, self.S, self.dev
        f = ops.render_forward(self.world, self.normals, self.h, self.M, self.V, self.znear, self.zfar, self.first, self.num,
                               self.colors, S, K, CUTOFF, THR, SIGMA, False, True, rows=p.rows, workspace_state=0)
        ws = _lib.clean_workspace(dev, ("render_forward_binned", self.N, self.P, S),
                                  lib.dss_render_forward_workspace(self.N, self.P, S, K))
        r0, r1, cyc = ops._band(p.rows, S)
        P_ = _lib.ptr
        valid, vis, img = f["valid"].view(torch.uint8), f["visible"].view(torch.uint8), f["image"]
        args = (P_(self.world), P_(self.normals), None, P_(self.h), None, None, P_(self.M), P_(self.V), P_(self.znear),
                P_(self.zfar), P_(self.first), P_(self.num), self.N, self.P, 1, 0, S, K, CUTOFF, SIGMA, THR, r0, r1, cyc,
                P_(self.colors), 3, P_(f["pts_screen"]), P_(f["ellipse_params"]), P_(f["radii"]), P_(f["scaler"]),
                P_(f["cutoff_threshold"]), P_(valid), P_(f["idx"]), P_(f["zbuf"]), P_(f["qvalue"]), P_(f["occupancy"]),
                P_(vis), P_(img), int(img.stride(0)), int(img.stride(1)), P_(f["wsum"]), P_(ws), ws.numel(), 2,
                _lib.stream_ptr(dev))
        run = lambda: lib.dss_render_forward(*args)
        _lib.check(run(), "dss_render_forward(DSS_WS_BINNED)")
        self._keep = (f, ws)
        mean, med = self._event_ms(run, iters)
        self.fine_single_ms, self.fine_batch_ms = mean, self._batch_ms(run)
        if self.fine_batch_ms is not None and self.fine_batch_ms < mean:
            return self.fine_batch_ms, min(med, self.fine_batch_ms)
        return mean, med

    def backward_gather_ms(self, iters=50):
        """The backward's dominant kernel (render_backward_kernel: blend backward + occupancy surrogate per visible
        point), timed on its own through the staged entry point dss_render_backward_gather (round 2 first subtracted the
        event time of dss_backward_radius from that of the full call: the stand-alone radius path is slower than the fused
        preparation, which flattered the gather by a third).  Also returns what the VALU roofline needs:
        the number of (pixel, visible point) pairs inside the search radius."""
        p, S = self.part, self.S
        f = ops.render_forward(self.world, self.normals, self.h, self.M, self.V, self.znear, self.zfar, self.first,
                               self.num, self.colors, S, K, CUTOFF, THR, SIGMA, False, True, rows=p.rows)
        # (multi-GPU path: the band gradient the last step's loss produced)
        g = p.slice(self.grad_out).contiguous() if self.grad_out is not None else self._st["g_band"]
        args = (g, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], f["visible"], self.first,
                self.num, RADII_S, CLIP)
        gf, gp, rs0 = ops.render_backward(*args, image_size=S, rows=p.rows, return_rs=True)
        full = lambda: ops.render_backward(*args, image_size=S, rows=p.rows, out=(gf, gp))
        # the gather kernel alone, on the lists / alpha plane / rs / zero-filled gradients the full call leaves behind
        gather = lambda: ops.render_backward(*args, image_size=S, rows=p.rows, out=(gf, gp), gather_only_rs=rs0)
        t_full, _ = self._event_ms(full, iters)
        full()
        t_gather, _ = self._event_ms(gather, iters)
        self.gather_single_ms, self.gather_batch_ms = t_gather, None
        try:
            # the same launch through a prebuilt ctypes call (the ops wrapper spends more host time per call than the kernel
            # runs): 20 back-to-back launches per event pair, like the fine kernel
            lib, P_ = _lib.load(), _lib.ptr
            r0, r1, cyc = ops._band(p.rows, S)
            ws = _lib.workspace(self.dev, lib.dss_render_backward_workspace(self.N, self.P, S))   # the buffer `full` filled
            vis8 = f["visible"].view(torch.uint8)
            gargs = (P_(g), P_(f["idx"]), P_(f["qvalue"]), P_(f["wsum"]), P_(f["scaler"]), P_(f["pts_screen"]),
                     P_(f["radii"]), P_(vis8), P_(self.first), P_(self.num), self.N, self.P, S, K, 3, r0, r1, cyc,
                     float(RADII_S), float(CLIP), P_(gf), P_(gp), P_(rs0), None, None, P_(ws), ws.numel(),
                     _lib.stream_ptr(self.dev))
            fast = lambda: lib.dss_render_backward_gather(*gargs)
            full()
            _lib.check(fast(), "dss_render_backward_gather")
            self.gather_batch_ms = self._batch_ms(fast)
            self._keep_gather = (g, f, gf, gp, rs0, ws, vis8)
            if 0.5 * t_gather < self.gather_batch_ms < t_gather:   # (a batch that is implausibly short did not run the kernel)
                t_gather = self.gather_batch_ms
        except Exception:  # noqa: BLE001  (keep the per-launch figure)
            torch.cuda.synchronize()
        t_prep = max(t_full - t_gather, 0.0)
        # (pixel, point) pairs the rule has to evaluate: pixel centres within rs of a visible, on-screen point
        rs = rs0
        vis = f["visible"]
        pts = f["pts_screen"][vis]
        cloud = (torch.arange(self.P, device=self.dev) // self.Pc)[vis]
        r = rs[cloud]
        ok = (pts[:, 2] >= 0) & (pts[:, 0].abs() <= 1) & (pts[:, 1].abs() <= 1)
        pts, r = pts[ok], r[ok]
        pairs = 0
        ndc = -1 + (2 * torch.arange(S, device=self.dev, dtype=torch.float32) + 1) / S
        own = torch.tensor(p.row_indices(), device=self.dev, dtype=torch.int64)   # image rows of this rank's band
        ys = ndc[S - 1 - own]                                  # their NDC rows (image row r <-> NDC index S-1-r)
        for i in range(0, pts.shape[0], 4096):                 # chunked: (points, S) masks per axis, exact disc count
            q, rr = pts[i:i + 4096], r[i:i + 4096]
            dx2 = (ndc[None, :] - q[:, 0:1]) ** 2
            dy2 = (ys[None, :] - q[:, 1:2]) ** 2
            # pixels with dx^2 + dy^2 <= rs^2: per row, the columns with dx^2 <= rs^2 - dy^2
            lim = (rr[:, None] ** 2 - dy2).clamp_min(-1.0)     # (pts, rows)
            dx2s, _ = dx2.sort(dim=1)
            pairs += int(torch.searchsorted(dx2s, lim.contiguous(), right=True).sum().item())
        return t_gather, t_full, t_prep, pairs, int(vis.sum().item())


def api_path_ms(wl, n=200, graphed=False, warm=400, calling_thread=False):
    """The same workload through the drop-in API a train_mvr.py user calls (DSS/core/renderer.py:36-82):
    `SurfaceSplattingRenderer(SurfaceSplatting(...), NormWeightedCompositor(), fused=True)(cloud)` + `.backward()`,
    eager, autograd and Python object handling included; h precomputed like the headline. -> ms per fwd+bwd
    (`graphed`: the renderer's graphed mode -- forward and backward replayed as two hipGraphs over static buffers; None if
    this build of the renderer has no such mode).
    This path is HOST-bound, and on the GPU boxes the host's speed for it varies by 2x between runs and between phases of one
    run (0.10 ... 0.20 ms per iteration for the same 58 us of GPU work; profiles/r4_c_api_path_variability.txt, tools/
    api_threads.py) -- so `warm` untimed iterations come first and the median of three blocks of `n` is reported."""
    import inspect
    from dss_amd.cloud import PointClouds3D
    from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
    dev = wl.dev
    R, T = look_at_view_transform(2.0, 30.0, [45.0 + 45.0 * k for k in range(wl.N)])
    cams = FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R, T=T, device=dev)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=CUTOFF, depth_merging_threshold=THR,
                                     Vrk_invariant=True, Vrk_isotropic=False, radii_backward_scaler=RADII_S,
                                     image_size=wl.S, points_per_pixel=K, bin_size=None, clip_pts_grad=CLIP,
                                     antialiasing_sigma=SIGMA)
    kw = {}
    if graphed:
        if "graphed" not in inspect.signature(SurfaceSplattingRenderer.__init__).parameters:
            return None
        kw["graphed"] = True
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                        fused=True, **kw)
    X = torch.nn.Parameter(wl.world.clone())
    C = torch.nn.Parameter(wl.colors[:wl.Pc].clone())
    h = wl.h[:1].clone()

    def step():
        X.grad = None
        C.grad = None
        img = renderer(PointClouds3D([X], [wl.normals], [C]), Vrk_h=h)
        img.backward(wl.grad_out)
    # calling_thread: the caller's explicit opt-in, scoped around the loop (dss_amd.calling_thread_backward); otherwise the
    # process's autograd state is whatever PyTorch's default is -- the renderer does not touch it
    import contextlib
    import dss_amd
    with (dss_amd.calling_thread_backward() if calling_thread else contextlib.nullcontext()):
        for _ in range(max(5, warm)):
            step()
        blocks = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t) / n * 1e3)
    return sorted(blocks)[1]


def cpu_baseline():
    """Reference CPU fallback (oracle/_ref = unmodified DSS/csrc/rasterize_points_cpu.cpp) timed on ONE
    host core (the code has no OpenMP / at::parallel_for) on the metric's own configuration: the same 32,684-point
    scene at 512x512, one forward + one backward (~30 s; BENCH_CPU_SIZE=256 selects the quarter-pixel sample of
    rounds 1-2, reported with its size)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import scenes
    pts, nrm, col, _ = bunny_cloud()
    h = scenes.global_h(pts)
    Sb = int(os.environ.get("BENCH_CPU_SIZE", str(S)))
    M, V, _ = scenes.camera_matrices(2.0, 30.0, 45.0)
    sc = scenes.setup_scene(pts, nrm, M, V, Sb, h=h)
    P = sc["points"].shape[0]
    rng = np.random.default_rng(1)
    gocc = rng.standard_normal((1, Sb, Sb)).astype(np.float32)
    R = oracle.ref()
    torch.set_num_threads(1)
    if R is not None:
        t = lambda k: torch.from_numpy(np.ascontiguousarray(sc[k]))
        t0 = time.perf_counter()
        R.splat_points(t("points"), t("ellipse"), t("cutoff"), t("radii"), t("first_idx"), t("num_pts"), THR, Sb, K, 0, 0)
        t_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        R._splat_points_occ_backward(t("points"), t("radii"), torch.from_numpy(gocc), t("first_idx"), t("num_pts"),
                                     RADII_S, THR)
        t_b = time.perf_counter() - t0
        kind = "reference"
        what = "reference CPU splat_points(bin_size=0) + _splat_points_occ_backward"
    else:
        t0 = time.perf_counter()
        o = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                 sc["num_pts"], Sb, K, THR, brute=True)
        t_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        oracle.splat_backward(sc["points"], sc["radii"], o[0], gocc, None, sc["first_idx"], sc["num_pts"], RADII_S)
        t_b = time.perf_counter() - t0
        kind = "port"
        what = "oracle C port (brute-force forward + fast backward)"
    return {"value": round(P / (t_f + t_b) / 1e6, 6), "unit": "Msplats/s", "cores": 1, "kind": kind,
            "sample": "%s, raster fwd+bwd only (no blend), same %d-point bunny scene at %dx%d (%s), one pass: fwd %.2fs + "
                      "bwd %.2fs on 1 of %d host cores"
                      % (what, P, Sb, Sb, "the metric's configuration" if Sb == S else "%.2f of the pixels of the %dx%d "
                         "workload" % (Sb * Sb / float(S * S), S, S), t_f, t_b, os.cpu_count())}


def self_launch(n_gpus: int) -> int:
    """Re-run this command line as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`
    (one process per GPU, rendezvous on 127.0.0.1 at a free port).  Returns the launcher's exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n_gpus and os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) are visible" % (n_gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["graph", "eager"], default=None,
                    help="graph: replay the captured step as a hipGraph (default on 1 GPU); eager: plain launches")
    ap.add_argument("--workload", choices=["cfg2", "cfg4", "cfg5"], default="cfg2",
                    help="cfg2 (default, the metric's configuration): BASELINE configs[1], one camera per rank (weak scaling). "
                         "cfg4 / cfg5: BASELINE configs[3] (1M points x 8 cameras @1024^2) / configs[4] (4M points @2048^2), "
                         "image rows sharded over the ranks (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiler passes: run warm-up + the timed region only (no per-kernel event timing, no API / kNN legs), "
                         "print a short JSON line")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not spawn the rocprofv3 counter passes (roofline.traffic is then null, traffic_source 'not measured')")
    args = ap.parse_args()
    large = args.workload != "cfg2"
    if args.steps is None:
        args.steps = 32 if large else 200   # (large: two periods of the cached point order, BENCH_ORDER_REFRESH = 16)
    if args.warmup is None:
        args.warmup = 5 if large else 20

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU under torch.distributed.run, the same
        # command line the driver would use), forward its output and exit code
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start one rank per GPU, or run without a launcher)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("BENCH_BACKWARD_FUSED"):   # development A/B: DSS_OPT_BACKWARD_FUSED (include/dss_hip.h)
        _lib.set_option(_lib.OPT_BACKWARD_FUSED, int(os.environ["BENCH_BACKWARD_FUSED"]))
    if os.environ.get("BENCH_BACKWARD_TPW"):     # development A/B: DSS_OPT_BACKWARD_TPW
        _lib.set_option(_lib.OPT_BACKWARD_TPW, int(os.environ["BENCH_BACKWARD_TPW"]))
    local = local % torch.cuda.device_count()  # (BENCH_DIST_BACKEND=gloo lets two ranks share one GPU in tests)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # BENCH_FORCE_DIST=1: take the multi-GPU code path -- process group (RCCL), second communicator, asynchronous image
    # all-gather, visibility all-gather, bucketed gradient all-reduce, graph segments beside the watchdog -- at ANY world
    # size, world size 1 included: a one-GPU box then executes every line a first 8-GPU run would
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"
    multi = world > 1 or force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        own_port = False
        if "MASTER_PORT" not in os.environ:   # (no launcher: forced world of one)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            own_port = True
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if rank == 0:
            print("bench.py: world %d, backend %s, %d visible device(s), torch %s, HSA_ENABLE_IPC_MODE_LEGACY=%s%s"
                  % (world, backend, torch.cuda.device_count(), torch.__version__,
                     os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), " (BENCH_FORCE_DIST)" if force_dist else ""),
                  file=sys.stderr, flush=True)
        for attempt in range(5):
            try:
                if backend == "nccl":
                    dist.init_process_group("nccl", device_id=dev)
                else:
                    dist.init_process_group(backend)
                break
            except Exception as e:  # noqa: BLE001
                # a forced world of one picks its own rendezvous port (bind to 0, close, listen): another process can take the
                # port in between (EADDRINUSE, seen once between two back-to-back runs) -- pick another one; a launcher's port
                # is the launcher's business
                if not own_port or attempt == 4 or "EADDRINUSE" not in str(e) and "address already in use" not in str(e):
                    raise
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # workload: cfg2 = one 32,684-point camera per rank at 512^2; cfg4 / cfg5 = a fixed job whose rows are shared
    cloud, S, cams = None, globals()["S"], world
    if large:
        cloud, S, cams = large_cloud(args.workload)
    # multi-GPU: tile-row-cyclic bands (rank g renders the 8-row tile rows g, g + G, ...: balanced for any scene, equal-size
    # all-gather) whenever the sizes allow it; BENCH_ROW_PARTITION=bands selects the contiguous equal bands of rounds 1-2
    # (round 5, emulated per-rank steps at 8 ranks, tools/band_timing.py: at the metric's configuration the cyclic bands take
    # 99-101 us on every rank against 91-112 for load-balanced contiguous ones; at configs[3] -- per-point work dominates, the
    # rows are cheap -- contiguous equal bands take 1.06-1.27 ms against 1.46-1.51 for the cyclic ones: the large workloads
    # default to contiguous bands)
    layout = os.environ.get("BENCH_ROW_PARTITION", "bands" if large else "cyclic")
    cyclic = world > 1 and world & (world - 1) == 0 and S % (8 * world) == 0 and layout == "cyclic"
    part = RowPartition(S, world, rank, cyclic=cyclic)
    wl = Workload(dev, cams, part, cloud=cloud, multi=multi)
    # end-of-forward exchange: BENCH_EXCHANGE=overlap (default) | fold | auto (both forms are prepared in the launch mode of the
    # timed region and timed on THESE ranks, max over the ranks, and the faster is kept): three collectives with the image bands
    # off the critical path, or two with the flags folded into the image all-gather
    exchange_note = None
    # (default: the three-collective form.  Measured at world size 1 in the launch mode of the timed region the folded form is
    # 11 us per step SLOWER -- its two extra kernels and the stream fork cost more than the collective they save,
    # profiles/r5_d_bench_forced_dist_world1_auto.json -- and at N > 1 it puts the image bytes on the critical path; the
    # comparison on the ranks of the run is BENCH_EXCHANGE=auto, and costs two more graph captures per rank)
    want_exchange = os.environ.get("BENCH_EXCHANGE", "overlap") if multi else None

    def capture(unroll=1, side=None):
        side = side or torch.cuda.Stream()   # (the forward workspace -- and a saved point order in it -- is cached per stream)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                wl.step()
        torch.cuda.current_stream().wait_stream(side)
        drain_collective_watchdog()
        g = torch.cuda.CUDAGraph()
        # capture on the stream the warm-up ran on: the zero-initialised forward workspace is cached per stream, and
        # a first use on a fresh capture stream would record its one-off torch.zeros fill (5.5 us) into every replay
        with torch.cuda.graph(g, stream=side):
            for _ in range(unroll):
                wl.step()
        return g

    def quick(fn, n=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    # launch mechanism: plain eager launches, one hipGraph replay per step, or UNROLL consecutive steps captured in one
    # graph (every step is the full, identical launch sequence; between two graph launches the stream idles for ~5-9 us,
    # which a training loop captured as a whole would not pay per iteration either).  Unless forced with --mode, a short
    # untimed calibration picks the fastest on a single GPU (all are recorded in `config`); multi-GPU runs are eager
    # (the RCCL calls stay outside any graph).
    UNROLL = 10
    graph, graph_u, ms_modes = None, None, {}
    unrollable = args.steps % UNROLL == 0 and args.steps >= UNROLL
    mode = args.mode or ("eager" if multi else None)
    order_graphs = None
    if large and wl.order_refresh > 0 and mode is None and not multi:
        # cached point order: a captured step freezes ONE of the two kinds of call, so both are captured -- the step that sorts
        # and saves the order, and the step that reuses it -- and replayed in the renderer's own rhythm (every k-th step saves)
        k_ord = wl.order_refresh
        ms_eager = quick(wl.step, n=k_ord)
        order_side = torch.cuda.Stream()
        wl.force_order = "save"
        g_save = capture(side=order_side)
        wl.force_order = "reuse"
        g_reuse = capture(side=order_side)
        wl.force_order = None
        tick = [0]

        def run_order_graphs():
            (g_save if tick[0] % k_ord == 0 else g_reuse).replay()
            tick[0] += 1
        ms_graphs = quick(run_order_graphs, n=k_ord)
        ms_modes = {"eager": ms_eager, "graph_save_reuse": ms_graphs}
        mode = min(ms_modes, key=ms_modes.get)
        order_graphs = run_order_graphs if mode == "graph_save_reuse" else None
    elif large and wl.order_refresh > 0 and mode is None:
        mode = "eager"
    def multi_runner():
        """the launch mechanism of the multi-GPU step for the CURRENT exchange form: graphs of the compute segments around
        host-issued collectives, and -- if RCCL lets itself be captured and the replay reproduces the eager step on every rank --
        the whole step as one graph.  -> (mode, run, steps per launch, segment note, whole-step note)"""
        mode, seg_note, whole_note = "eager", None, None
        if True:
            # multi-GPU: (i) the compute segments between the RCCL calls as graphs; (ii) if RCCL lets itself be captured, the whole
            # step -- launches and collectives -- as ONE graph, checked against the eager step on every rank before it is used
            try:
                barrier()   # (no collective in flight while capturing)
                wl.capture_segments()
                mode = "graph_segments"
            except Exception as e:  # noqa: BLE001  (capture refused: plain launches, and say so)
                seg_note = "segment capture failed: %s: %s" % (type(e).__name__, str(e)[:160])
                mode = "eager"
            whole_wanted = os.environ.get("BENCH_NO_WHOLE_GRAPH") != "1" and dist.get_backend() == "nccl"
            if not whole_wanted:
                # (a host-side backend -- gloo in the CPU-collective tests -- cannot be captured, and a refused capture leaves the
                # thread's capture state unusable: not attempted)
                whole_note = "not attempted (backend %s%s)" % (dist.get_backend(), ", BENCH_NO_WHOLE_GRAPH" if os.environ.get("BENCH_NO_WHOLE_GRAPH") == "1" else "")
            if whole_wanted:
                ok = 1.0
                try:
                    barrier()
                    ref = [t.clone() for t in wl.step()]
                    barrier()
                    got = wl.capture_whole_step()
                    got = wl.step_whole()
                    torch.cuda.synchronize()
                    same = torch.equal(got[0], ref[0]) and all(
                        float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-30) for a, b in zip(got[1:], ref[1:]))
                    if not same:
                        ok, whole_note = 0.0, "whole-step graph gave other results than the eager step: not used"
                except Exception as e:  # noqa: BLE001  (capture of the collectives refused: the segments stay)
                    ok, whole_note = 0.0, "whole-step capture failed: %s: %s" % (type(e).__name__, str(e)[:160])
                flag = torch.tensor([ok], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # (every rank takes the same path)
                if float(flag.item()) == 1.0:
                    mode = "graph_step"
                    if unrollable and os.environ.get("BENCH_NO_WHOLE_GRAPH_UNROLL") != "1":
                        # ten consecutive steps per graph launch, as on one GPU (the stream idles ~8 us between two graph
                        # launches): same captured step, checked the same way
                        try:
                            one = wl._whole
                            got = wl.capture_whole_step(UNROLL)
                            got = wl.step_whole()
                            torch.cuda.synchronize()
                            same = torch.equal(got[0], ref[0]) and all(
                                float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-30) for a, b in zip(got[1:], ref[1:]))
                            oku = 1.0 if same else 0.0
                        except Exception:  # noqa: BLE001
                            oku = 0.0
                        flag = torch.tensor([oku], device=dev)
                        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                        if float(flag.item()) == 1.0:
                            mode = "graph_step_x%d" % UNROLL
                        else:
                            wl._whole = one
                elif whole_note is None:
                    whole_note = "another rank could not capture the whole step"
        spl = UNROLL if mode.startswith("graph_step_x") else 1
        run = wl.step_whole if mode.startswith("graph_step") else (wl.step_segments if mode == "graph_segments" else wl.step)
        return mode, run, spl, seg_note, whole_note

    seg_note = None
    whole_note = None
    if multi and args.mode != "eager":
        # end-of-forward exchange: measured, not assumed (BENCH_EXCHANGE=overlap|fold fixes it).  Every form is prepared in the
        # launch mechanism the timed region will use and timed there (20 launches, max over the ranks: every rank takes the
        # same decision); the faster one is prepared again and kept.  (Eager steps would mislead: at world size 1 the folded
        # form is faster eagerly -- fewer host calls -- and slower as a graph -- two more kernels and a stream fork.)
        forms = [want_exchange] if want_exchange in ("overlap", "fold") else ["overlap", "fold"]
        t_form = {}
        for form in forms:
            try:
                wl.set_exchange(form == "fold")
                m_, run_, spl_, _, _ = multi_runner()
                for _ in range(3):
                    run_()
                barrier()
                t0 = time.perf_counter()
                for _ in range(20):
                    run_()
                torch.cuda.synchronize()
                tt = torch.tensor([(time.perf_counter() - t0) / (20 * spl_) * 1e3], device=dev, dtype=torch.float64)
            except Exception as e:  # noqa: BLE001  (a form that cannot run here drops out of the comparison)
                tt = torch.tensor([float("inf")], device=dev, dtype=torch.float64)
                exchange_note = {"error_" + form: "%s: %s" % (type(e).__name__, str(e)[:160])}
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_form[form] = float(tt.item())
        best = min(t_form, key=t_form.get)
        note = dict(exchange_note or {})
        note.update({"form": best, "chosen": "BENCH_EXCHANGE" if len(forms) == 1 else
                     "measured: 20 launches of the prepared step per form, max over the ranks",
                     "ms_per_step": {k: (round(v, 5) if v != float("inf") else None) for k, v in t_form.items()}})
        exchange_note = note
        wl.set_exchange(best == "fold")
        mode, _run, _spl, seg_note, whole_note = multi_runner()
    elif multi:
        wl.set_exchange(want_exchange == "fold")
        exchange_note = {"form": "fold" if want_exchange == "fold" else "overlap", "chosen": "eager launches: not compared"}
    if mode is None:
        graph = capture()
        ms_modes = {"eager": quick(wl.step, n=8 if large else 40), "graph": quick(graph.replay, n=8 if large else 40)}
        if unrollable and not large:
            graph_u = capture(UNROLL)
            ms_modes["graph_x%d" % UNROLL] = quick(graph_u.replay, n=8) / UNROLL
        mode = min(ms_modes, key=ms_modes.get)
    elif mode == "graph":
        graph = capture()
    steps_per_launch = UNROLL if (mode.startswith("graph_x") or mode.startswith("graph_step_x")) else 1
    if mode.startswith("graph_step"):
        run = wl.step_whole
    else:
        run = graph_u.replay if steps_per_launch > 1 else (graph.replay if mode == "graph" else
                                                             (wl.step_segments if mode == "graph_segments" else wl.step))
    if order_graphs is not None:
        run = order_graphs

    for _ in range(-(-args.warmup // steps_per_launch)):
        run()

    def timed_block():
        """exactly args.steps steps between barrier + synchronize on both sides; max over ranks"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps // steps_per_launch):   # (steps_per_launch divides args.steps)
            run()
        barrier()
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt
    # A single block of K steps is 1.6 ms at the default sizes of round 2: one scheduler hiccup moved the headline by
    # several percent.  The block of EXACTLY K steps is therefore repeated until at least MIN_TIMED_S of timed work has
    # accumulated (every rank takes the same decision from the all-reduced times); the reported step time is the MEDIAN
    # block, min / max / count are recorded in `config.timing`.
    MIN_TIMED_S, MAX_BLOCKS = 0.2, 400
    blocks = [timed_block()]
    while sum(blocks) < MIN_TIMED_S and len(blocks) < MAX_BLOCKS:
        blocks.append(timed_block())
    dt = sorted(blocks)[len(blocks) // 2]
    ms_step = dt / args.steps * 1e3
    splats = wl.P  # cameras * points per cloud submitted per step (whole job)
    value = splats / (ms_step * 1e-3) / 1e6

    if args.timed_only:
        if rank == 0:
            print(json.dumps({"metric": "Msplats/s fwd+bwd @%d^2" % S, "value": round(value, 3), "ms_per_step": round(ms_step, 5),
                              "launch": mode, "timed_only": True}))
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

    # second reported figure (single GPU): the same step with the variance-scale statistic h recomputed inside it -- the
    # kNN-7 of rasterizer.py:310-326, which the reference reruns every iteration (refresh=True default, :293, :344).  The
    # headline `value` takes h as an input of the step (SURVEY section 8 files the kNN under "next"); both are reported.
    value_knn = ms_knn = knn_mode = None
    if not multi and not large:
        one = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt = torch.full((1,), wl.Pc, dtype=torch.int64, device=dev)

        def step_with_knn():
            h = ops.cloud_mean_clamp(ops.knn_kth_sqdist(wl.world, one, cnt, 7), one, cnt, 0.5, 5e-5, 1e-3, 0.5e-3, 7)
            wl.h = h.expand(wl.N).contiguous() if wl.N > 1 else h
            return wl.step()
        ms_knn, knn_mode = quick(step_with_knn, n=max(20, args.steps // 4)), "eager"
        ms_knn_modes = {"eager": ms_knn}
        if mode != "eager":
            # the same launch mechanism as the headline: the kNN chain + the step captured in one hipGraph (h is written
            # into the buffer the captured step reads)
            try:
                h_buf = wl.h

                def step_with_knn_static():
                    h = ops.cloud_mean_clamp(ops.knn_kth_sqdist(wl.world, one, cnt, 7), one, cnt, 0.5, 5e-5, 1e-3, 0.5e-3, 7)
                    h_buf.copy_(h.expand_as(h_buf))
                    return wl.step()
                wl.h = h_buf
                keep_step, wl_step = wl.step, step_with_knn_static
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        wl_step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, stream=side):
                    for _ in range(steps_per_launch):
                        wl_step()
                ms_g = quick(gk.replay, n=max(8, args.steps // (4 * steps_per_launch))) / steps_per_launch
                ms_knn_modes[mode] = ms_g
                if ms_g < ms_knn:
                    ms_knn, knn_mode = ms_g, mode
            except Exception as e:  # noqa: BLE001  (capture refused: keep the eager figure)
                knn_mode = "eager (graph capture failed: %s)" % type(e).__name__
        value_knn = splats / (ms_knn * 1e-3) / 1e6
    # third reported figure (single GPU): the CAUSAL form of the step -- the one `bench.py --gpus N` times on N > 1 ranks: the
    # image gradient is not an input but the reference's image loss (Trainer.calc_dr_loss) of the rendered image, computed
    # inside the step.  The like-for-like single-GPU line for a scaling ratio against the N > 1 values.
    value_loss = ms_loss = loss_mode = None
    if not multi and not large:
        try:
            wl_c = Workload(dev, 1, part, multi="local")
            ms_loss, loss_mode = quick(wl_c.step, n=max(20, args.steps // 4)), "eager"
            if mode != "eager":
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        wl_c.step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gl = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gl, stream=side):
                    for _ in range(steps_per_launch):
                        wl_c.step()
                ms_g = quick(gl.replay, n=max(8, args.steps // (4 * steps_per_launch))) / steps_per_launch
                if ms_g < ms_loss:
                    ms_loss, loss_mode = ms_g, mode
            value_loss = splats / (ms_loss * 1e-3) / 1e6
        except Exception as e:  # noqa: BLE001  (an extra figure must not take the headline down)
            loss_mode = "failed: %s: %s" % (type(e).__name__, str(e)[:120])
    api = not multi and not large
    ms_api = api_path_ms(wl) if api else None                                                # PyTorch's default autograd state
    ms_api_ct = api_path_ms(wl, calling_thread=True) if api else None                        # the caller's scoped opt-in
    ms_api_graphed = api_path_ms(wl, graphed=True) if api else None
    ms_api_graphed_ct = api_path_ms(wl, graphed=True, calling_thread=True) if api else None

    dist_block = {"world_size": 1, "backend": None}
    if multi:
        # diagnosable multi-GPU line: per-rank compute min / max, time in each collective, the
        # communicator set-up that was actually used, library versions
        tm = wl.dist_timing()
        keys = sorted(tm)
        mine = torch.tensor([tm[k] for k in keys], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        allt = torch.stack(allt).cpu()
        nccl_v = None
        try:
            nccl_v = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            pass
        eng = wl.engine
        stage_list = [("%s:%s" % (k, l)) for k, l, _ in wl.stages()]
        n_coll = sum(1 for k, _, _ in wl.stages() if k == "x")
        dist_block = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "forced": force_dist,
                      "rccl_version": nccl_v,
                      "visible_devices": torch.cuda.device_count(), "partition": part.describe(),
                      "engine": "dss_amd.sharded.RowShardedRender (the object SurfaceSplattingRenderer(row_partition=...) drives)",
                      "causal": True,
                      "loss": "Trainer.calc_dr_loss (trainer.py:332-376) of the rank's own band against fixed targets (the scene's "
                              "render shifted by (5, 9) pixels), block partials of its five per-image sums all-reduced (%d bytes); two launches: "
                              "dss_image_loss_band_partials | dss_image_loss_band_backward_partials" % (2560 * wl.N),
                      "stages": stage_list,
                      "exchange": exchange_note, "collectives_per_step": n_coll,
                      "collectives_on_the_critical_path": n_coll - (0 if eng.fx.fold else 1),
                      "gradient_exchange": ("owner: the rank of a point's centre row computes the pair's whole position gradient "
                                            "(dss_render_backward_owned_plane) from the all-gathered alpha-gradient plane (%d bytes "
                                            "per rank), clip + projection, then ONE all-reduce of %d bytes"
                                            % (wl.N * part.band * S * 4, wl.Pc * 24)) if wl.owner else
                                           ("bucket: partial sums of every (camera, point) pair from the band's own gradient, ONE "
                                            "all-reduce of %d bytes, clip + projection behind it" % (wl.P * 24)),
                      "defaults": "provisional: chosen from single-GPU emulation of the ranks and RCCL at world size 1; no N > 1 "
                                  "RCCL run has confirmed them (BENCH_GRADIENT / BENCH_ROW_PARTITION / BENCH_EXCHANGE override)",
                      "overlap": bool(eng.fx.overlap) and not eng.fx.fold,
                      "image_issue": "behind the backward (side stream)" if eng.late_image else "at the end of the forward, asynchronous",
                      "degraded": eng.fx.degraded, "segment_capture": seg_note or "ok",
                      "whole_step_graph": whole_note or "ok",
                      "timing_us": {k: {"min": round(float(allt[:, i].min()), 1), "max": round(float(allt[:, i].max()), 1),
                                        "mean": round(float(allt[:, i].mean()), 1)} for i, k in enumerate(keys)},
                      "timing_how": "HIP events on the compute stream around each stage of 20 eager steps, per rank; "
                                    "min / max / mean over the ranks"}

    # ---- roofline of the dominant kernel, picked from a per-kernel event-timing pass --------------------------------
    kit = 10 if large else 50
    fine_mean, fine_med = wl.fine_kernel_ms(iters=kit)
    gather_ms, bwd_ms, prep_ms, pairs, n_vis = wl.backward_gather_ms(iters=kit)
    r0, r1 = 0, part.n_rows
    # HBM: algorithmic bytes of ONE fine-kernel launch: every pixel of the band writes idx+zbuf+qvalue
    # (12K B) + occ (4 B) + RGBA (16 B) + wsum (4 B); every splat's screen record (pos 12, ellipse 12, radii 8, cutoff 4) +
    # scaler (4) + colour (12) = 52 B is read once.
    alg_bytes = wl.N * (r1 - r0) * S * (12 * K + 4 + 16 + 4) + wl.P * 52
    achieved = alg_bytes / (fine_mean * 1e-3) / 1e9
    traffic, traffic_src, gather_traffic, prof_ms = None, None, None, {}
    if not multi and rank == 0 and not args.no_traffic:
        # HBM bytes per launch are PMC counters: they cannot be read in-process.  tools/collect_traffic.py runs this very
        # command (eager, 20 steps, --no-traffic) twice under `rocprofv3 --kernel-trace --pmc` (FETCH_SIZE and WRITE_SIZE in
        # separate passes, no other trace domain) and averages them over the fine_kernel dispatches: measured in THIS run.
        # The same two passes carry the kernel trace, i.e. the rocprofv3 average duration of both roofline kernels.
        import shutil
        import subprocess
        if shutil.which("rocprofv3"):
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "collect_traffic.py")] +
                                   (["--workload", args.workload] if large else []), capture_output=True, text=True,
                                   timeout=900 if large else 240)
                tj = json.loads(r.stdout.strip().splitlines()[-1])
                traffic = int(tj["traffic_bytes_per_launch"])
                gather_traffic = tj.get("render_backward_kernel_traffic_bytes_per_launch")
                prof_ms = tj.get("rocprof_kernel_ms") or {}
                traffic_src = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over the "
                               "same step, FETCH x2 (gfx950), %d + %d fine_kernel dispatches" % tuple(tj["samples"]))
            except Exception as e:  # noqa: BLE001  (no profiler on the box, counters unavailable, timeout)
                traffic_src = "live measurement failed (%s); " % type(e).__name__
    if traffic is None:
        # no committed fallback any more (rounds 2-5 quoted a stale file here): what this run did not measure it does not report
        traffic_src = (traffic_src or "") + "not measured"
    # the fractions are computed from the rocprofv3 average of the kernel whenever this run measured it
    # (`kernel_ms_rocprofv3`); the event timings stay in the line beside it
    fine_prof = prof_ms.get("fine_kernel")
    fine_t = fine_prof if fine_prof else fine_mean
    achieved = alg_bytes / (fine_t * 1e-3) / 1e9
    hbm = {"bound": "hbm", "kernel": "fine_kernel<5> (fine pass + fused blend)", "achieved": round(achieved, 2),
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
           "frac_hbm": round(achieved / HBM_PEAK_GBS, 5), "frac_valu": None, "traffic": traffic,
           "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes,
           "kernel_ms_rocprofv3": None if fine_prof is None else round(fine_prof, 5),
           "kernel_ms_mean": round(fine_mean, 5),
           "kernel_ms_median": round(fine_med, 5), "kernel_ms_single_events": round(wl.fine_single_ms, 5),
           "kernel_ms_batch_of_20": round(wl.fine_batch_ms, 5),
           "frac_from": "kernel_ms_rocprofv3" if fine_prof else "kernel_ms_mean",
           "how": "achieved / frac use the rocprofv3 --kernel-trace average of the kernel measured in this run when available "
                  "(kernel_ms_rocprofv3), else HIP events on the launch stream: the smaller of (a) one event pair per launch, "
                  "mean of 50, and (b) one event pair around 20 back-to-back launches, / 20 (the event pair itself costs "
                  "3-4 us on a 17 us kernel)"}
    # VALU: the backward gather evaluates the occupancy rule of rasterize_points_backward.cu:141-178 for every (pixel,
    # visible point) pair inside the search radius: dx, dy, d2 (fma), two range compares, the g>0 / bbox skip (3), max,
    # rcp, the product with g and two accumulating fmas = MIN_OPS lane operations.  Peak = 256 CUs x 4 SIMDs x 32 lanes x
    # 2.4 GHz lane operations per second (= the 157.3 TFLOP/s fp32 vector peak of MI355X_MICROARCH.md with an FMA as two).
    MIN_OPS, VALU_PEAK = 12, 256 * 4 * 32 * 2.4e9 / 1e12
    gather_prof = prof_ms.get("render_backward_kernel")
    gather_t = gather_prof if gather_prof else gather_ms
    valu_ach = pairs * MIN_OPS / (gather_t * 1e-3) / 1e12 if gather_t > 0 else 0.0
    # the same kernel read against HBM with SURVEY 8(d)'s backward bytes: N S^2 (16 + 4 + 12K) [grad RGBA, grad_occ,
    # fragments re-read] + N P (48 + 24) [record re-read, gradients written]
    bwd_alg_bytes = wl.N * (r1 - r0) * S * (16 + 4 + 12 * K) + wl.P * 72
    bwd_hbm = bwd_alg_bytes / (gather_t * 1e-3) / 1e9 if gather_t > 0 else 0.0
    valu = {"bound": "valu", "kernel": "render_backward_kernel<3> (blend backward + occupancy surrogate per visible point)",
            "achieved": round(valu_ach, 4), "peak": round(VALU_PEAK, 2), "unit": "Tlaneop/s", "frac": round(valu_ach / VALU_PEAK, 5),
            "frac_valu": round(valu_ach / VALU_PEAK, 5), "frac_hbm": round(bwd_hbm / HBM_PEAK_GBS, 5),
            "algorithmic_bytes": bwd_alg_bytes, "achieved_hbm_GBps": round(bwd_hbm, 2),
            "pairs": pairs, "min_ops_per_pair": MIN_OPS, "visible_points": n_vis,
            "kernel_ms_rocprofv3": None if gather_prof is None else round(gather_prof, 5),
            "kernel_ms_mean": round(gather_ms, 5),
            "kernel_ms_single_events": round(wl.gather_single_ms, 5),
            "kernel_ms_batch_of_20": None if wl.gather_batch_ms is None else round(wl.gather_batch_ms, 5),
            "frac_from": "kernel_ms_rocprofv3" if gather_prof else "kernel_ms_mean",
            "how": "fractions from the rocprofv3 average of the kernel in the step when this run measured it, else HIP events "
                   "around dss_render_backward_gather alone: the smaller of one pair per launch (mean of 50) and "
                   "one pair around 20 back-to-back launches / 20 (a pair adds 3-4 us to a 20 us kernel) (whole "
                   "dss_render_backward: %.5f ms, i.e. %.5f ms outside the gather stage)" % (bwd_ms, prep_ms),
            "traffic": gather_traffic,
            "traffic_source": traffic_src if gather_traffic is not None else None}
    dominant, other = (valu, hbm) if gather_t > fine_t else (hbm, valu)
    if rank == 0:
        if large:
            wtxt = ("BASELINE configs[%d]: synthetic %d-point cloud, %d camera(s), %dx%d, K=5, fwd+bwd, image rows sharded "
                    "over %d rank(s), randomly ordered points, density-scaled h; point order of the binning %s"
                    % (3 if args.workload == "cfg4" else 4, wl.Pc, wl.N, S, S, world,
                       ("cached by the renderer: sorted and saved every %d-th step, reused by the others (all inside the "
                        "timed region)" % wl.order_refresh) if wl.order_refresh > 0 else "sorted in every step"))
        else:
            wtxt = ("BASELINE configs[1]: bunny-8000 x4 jitter = %d pts/cloud, %d camera(s), "
                    "512x512, K=5, fwd+bwd (setup+raster+blend and their backward), "
                    "%s, variance scale h precomputed (kNN-7 outside the step; "
                    "see value_with_knn)" % (wl.Pc, wl.N, "image gradient = the reference's image loss of the rank's own rows "
                                             "(inside the step, sums all-reduced)" if multi else "grad_out=randn(seed 1)"))
        rec = {
            "metric": "Msplats/s fwd+bwd @%d^2" % S, "value": round(value, 3), "unit": "Msplats/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 5),
            "higher_is_better": True, "scaling": "strong" if large else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wtxt,
                       "points_per_cloud": wl.Pc, "cameras": wl.N, "image_size": S, "points_per_pixel": K,
                       "h_precomputed": True, "parallelism": "rows%d" % world, "launch": mode,
                       "steps_per_graph_launch": steps_per_launch,
                       "timing": {"blocks_of_K_steps": len(blocks), "reported": "median block",
                                  "ms_per_step_min": round(min(blocks) / args.steps * 1e3, 5),
                                  "ms_per_step_max": round(max(blocks) / args.steps * 1e3, 5),
                                  "timed_seconds": round(sum(blocks), 4)},
                       "dist": dist_block},
            "roofline": dominant, "roofline_other": other,
        }
        if value_knn is not None:
            rec["value_with_knn"] = round(value_knn, 3)
            rec["ms_per_step_with_knn"] = round(ms_knn, 5)
            rec["with_knn_launch"] = knn_mode
            rec["knn_chain_ms"] = round(ms_knn - ms_step, 5)
            rec["ms_per_step_with_knn_by_launch"] = {k: round(v, 5) for k, v in ms_knn_modes.items()}   # (the chain is device-bound)
            rec["with_knn"] = ("the same step with the kNN-7 statistic of rasterizer.py:310-326 recomputed inside it (the reference's "
                               "refresh=True default): six launches (bounding box | cell counts + grid | scan | fill | query | mean)")
        if loss_mode is not None:
            rec["value_with_image_loss"] = None if value_loss is None else round(value_loss, 3)
            rec["ms_per_step_with_image_loss"] = None if ms_loss is None else round(ms_loss, 5)
            rec["with_image_loss_launch"] = loss_mode
            rec["with_image_loss"] = ("the causal form of the step (what --gpus N times on N > 1 ranks): render -> "
                                      "Trainer.calc_dr_loss of the rendered image against fixed targets -> backward, one rank, "
                                      "no collective (Workload(multi='local'))")
        if ms_api is not None:
            to_v = lambda ms: round(splats / (ms * 1e-3) / 1e6, 3)
            rec["value_via_api"] = to_v(ms_api)
            rec["ms_per_step_via_api"] = round(ms_api, 5)
            rec["via_api"] = ("SurfaceSplattingRenderer(fused=True)(cloud) + .backward(), eager, autograd included, h precomputed; "
                              "median of 3 x 200 iterations after 400 untimed ones (host-bound path).  value_via_api: PyTorch's "
                              "default autograd state (backward handed to the engine's device thread; constructing a renderer "
                              "does not change it).  *_calling_thread: the loop inside `with dss_amd.calling_thread_backward():` "
                              "(the caller's scoped opt-in).  *_graphed: SurfaceSplattingRenderer(graphed=True), forward and "
                              "backward replayed as two hipGraphs over static buffers")
            rec["value_via_api_calling_thread"] = to_v(ms_api_ct)
            rec["ms_per_step_via_api_calling_thread"] = round(ms_api_ct, 5)
            if ms_api_graphed is not None:
                rec["value_via_api_graphed"] = to_v(ms_api_graphed)
                rec["ms_per_step_via_api_graphed"] = round(ms_api_graphed, 5)
                rec["value_via_api_graphed_calling_thread"] = to_v(ms_api_graphed_ct)
                rec["ms_per_step_via_api_graphed_calling_thread"] = round(ms_api_graphed_ct, 5)
        for k, v in ms_modes.items():
            rec["config"]["calibration_ms_per_step_" + k] = round(v, 5)
        if not args.no_cpu_baseline and not large and not force_dist and world == 1:   # (rank 0 at N = 1 only: 30 s of one host core)
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
