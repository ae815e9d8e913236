"""Row-sharded render step: ONE rank's share of a forward + backward of the fused renderer whose image rows are partitioned
over the ranks of a process group (one process per GPU; ``nccl`` = RCCL over xGMI, ``gloo`` in the CPU tests).

The reference has no distributed layer (``rasterizer.py:236-240`` is its only batching); the seam this sits
behind is ``SurfaceSplattingRenderer.forward`` (``DSS/core/renderer.py:36-82``) -- `dss_amd.renderer` hands the call to
`RowShardedRender` when it was given a row partition, `bench.py --gpus N` drives the same object directly.

A step is CAUSAL: the gradient the backward consumes is computed from what the rank really holds at that moment.  Two ways
for a training loop to use it, both behind the drop-in classes:

* **replicated loss** (an unmodified loop, e.g. the reference's ``Trainer.compute_loss``, ``trainer.py:275-376``): the forward
  returns the FULL image on every rank -- the all-gather of the RGBA bands completes before the forward returns -- every rank
  evaluates the same loss on it, and the backward receives the full image gradient: the rank differentiates its own rows;
* **band loss** (`dss_amd.distributed.band_image_loss`: ``Trainer.calc_dr_loss`` with its per-image sums all-reduced): the
  forward returns the rank's band only; the image all-gather runs asynchronously for whoever wants to look at the picture.

Gradient exchange, either way (``gradient=``):

* ``"owner"``: the occupancy surrogate of a (camera, point) pair -- its whole search window, over all image rows -- is
  computed by the rank whose band holds the image row of the point's centre (``dss_render_backward_owned[_plane]``).  It needs
  the occupancy (alpha) gradient of ALL rows: with a replicated loss every rank has it; with a band loss the ranks all-gather
  that one channel (``N S^2 4`` bytes in total, `AlphaPlaneExchange`) in front of the backward.  A pair's position gradient is
  then complete on its owner, so clip + projection run first and ONE all-reduce carries the world-space sums;
* ``"bucket"``: every rank adds the pixels of its rows for every pair; ONE all-reduce of the per-pair partial sums
  (``P (3 + C)`` floats), clip + projection behind it.  Needs only the band's own gradient.

Both need the union of the visibility flags before the backward (``rs`` is the median radius of the GLOBALLY visible points,
``rasterizer.py:885-888``): an all-reduce (MAX) of P bytes issued at the end of the forward.
"""
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from .distributed import AlphaPlaneExchange, OverlappedExchange, RowPartition

__all__ = ["RowShardedRender", "default_partition", "choose_gradient_exchange"]


def default_partition(image_size: int, group=None, layout: str = "auto") -> Optional[RowPartition]:
    """The row partition of this rank in the initialised process group (None without one or in a world of one):
    tile-row-cyclic bands where the sizes allow it (``layout`` "auto" / "cyclic": balanced for any scene, equal-size
    exchange), else contiguous equal bands (``layout`` "bands")."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    G, g = dist.get_world_size(group), dist.get_rank(group)
    if G <= 1:
        return None
    S = int(image_size)
    cyclic = layout in ("auto", "cyclic") and G & (G - 1) == 0 and S % (8 * G) == 0
    if layout == "cyclic" and not cyclic:
        raise ValueError("a tile-row-cyclic partition needs a power-of-two world size and S %% (8 G) == 0, got G=%d S=%d" % (G, S))
    return RowPartition(S, G, g, cyclic=cyclic)


def choose_gradient_exchange(N: int, Pw: int, P: int, S: int, C: int, world_size: int, band_loss: bool = True,
                             features_shared: bool = False) -> str:
    """"owner" or "bucket" from the bytes each form puts on the critical path of a step (PROVISIONAL: xGMI link peak of
    MI355X_MICROARCH.md at half efficiency, 10 us per extra collective; no multi-GPU run has calibrated it).  bucket: ONE
    all-reduce of P (3 + C) floats; owner: an all-reduce of the world-space sums plus -- with a band-local loss -- the
    all-gather of the alpha-gradient plane.  Small jobs (the metric's configuration: 6 MB of partial sums) take the bucket
    form, large ones (configs[3]: 192 MB against 24 + 4 MB) the owner form."""
    G = max(int(world_size), 2)
    bw = min(G - 1, 7) * 153.0e3 * 0.5                 # bytes per microsecond
    ar = lambda b: 2.0 * b * (G - 1) / G / bw
    bucket = ar(4.0 * P * (3 + C))
    nf = Pw if features_shared else P
    owner = ar(4.0 * (3 * Pw + nf * C)) + ((4.0 * N * S * S / G) * (G - 1) / bw + 10.0 if band_loss else 0.0)
    return "bucket" if bucket <= owner else "owner"


class RowShardedRender:
    """Buffers, exchanges and the launch sequence of one rank for one problem shape.  ``N`` cameras, ``Pw`` world points
    (= points of the shared cloud, or ``P`` for per-camera clouds), ``P`` packed points, ``S`` image side, ``K`` points
    per pixel, ``C`` feature channels.

    ``features_shared``: the packed features are one ``(Pw, C)`` tensor repeated per camera (colours of a cloud extended to
    its cameras): the feature gradient is then summed over the cameras in the projection launch and the reduction carries
    ``Pw (3 + C)`` floats instead of ``3 Pw + P C``.
    ``gather_image`` False: no image all-gather at all (a band loss that never looks at the full picture).
    ``static_buffers``: see the constructor.
    ``force``: issue every collective even in a world of one rank (bench.py BENCH_FORCE_DIST: a one-GPU box then executes the
    RCCL code path line for line)."""

    def __init__(self, part: RowPartition, N: int, Pw: int, P: int, S: int, K: int, C: int, device, shared: bool,
                 cutoff: float, sigma: float, thr: float, backface: bool = False, group=None, gradient: str = "owner",
                 features_shared: bool = False, fold: bool = False, force: bool = False, late_image: bool = False,
                 gather_image: bool = True, static_buffers: bool = True):
        if gradient == "auto":
            gradient = choose_gradient_exchange(N, Pw, P, S, C, part.world_size, True, features_shared and shared)
        if gradient not in ("owner", "bucket"):
            raise ValueError("gradient must be 'owner', 'bucket' or 'auto', got %r" % (gradient,))
        if part.S != S:
            raise ValueError("the partition is for %d rows, the image has %d" % (part.S, S))
        self.part, self.group, self.dev = part, group, torch.device(device)
        self.N, self.Pw, self.P, self.S, self.K, self.C = int(N), int(Pw), int(P), int(S), int(K), int(C)
        self.shared, self.owner = bool(shared), gradient == "owner"
        self.features_shared = bool(features_shared) and self.shared
        self.settings = (float(cutoff), float(thr), float(sigma), bool(backface))
        self.active = part.world_size > 1 or bool(force)          # False: a world of one without `force` issues no collective
        self.gather_image = bool(gather_image)
        # True: per-point outputs of the forward that the band leaves alone live in buffers of this object (zero-filled once;
        # ONE step in flight -- bench.py, a captured step).  False: every forward gets fresh zero-filled ones, so that the
        # tensors an autograd node keeps stay untouched by a later render (the drop-in classes).
        self.static_buffers = bool(static_buffers)
        dev = self.dev
        self._fx = {}
        self._fx_args = dict(force=bool(force))
        self.late_image = bool(late_image)
        self.set_exchange(bool(fold))
        f32 = dict(dtype=torch.float32, device=dev)
        # reduction buffers: "bucket" = [feature partials (P,C) | position partials (P,3)]; "owner" = [world-space position
        # sums (Pw,3) | feature sums ((Pw,C) when the features are shared, else (P,C))]
        self.bucket = torch.zeros(self.P * (self.C + 3), **f32)
        nf = self.Pw if self.features_shared else self.P
        self.wbucket = torch.zeros(self.Pw * 3 + nf * self.C, **f32)
        self._pts_scratch = torch.zeros((self.P, 3), **f32)       # owner, features not shared: screen-space gradients
        # the three per-point outputs DSS_WS_BAND_OUTPUTS leaves alone for splats outside the band: zero-filled once
        self._point_outputs = (torch.zeros((self.P, 3), **f32), torch.zeros((self.P,), **f32), torch.zeros((self.P,), **f32))
        self.alpha_x = None                                       # band loss + owner: built on first use
        self.f = None                                             # outputs of the last forward
        self.vis_all = None
        self._image_pending = False                               # an image all-gather nobody has waited for yet
        self._bwd = None                                          # state handed from stage to stage of a backward
        self.mark = None                                          # diagnostics: callable(label), called between the stages

    # -- exchanges ---------------------------------------------------------------------------------------------------
    def set_exchange(self, fold: bool):
        """end-of-forward exchange: `overlap` (visibility all-reduce + asynchronous image all-gather on a second
        communicator) or `fold` (the flags ride in one blocking image all-gather); see `OverlappedExchange`"""
        fold = bool(fold)
        if fold not in self._fx:
            self._fx[fold] = OverlappedExchange(self.part, self.N, self.C + 1, self.P, self.dev, group=self.group,
                                                fold=fold, **self._fx_args)
            self._fx[fold].late_image = self.late_image
        self.fx = self._fx[fold]

    @property
    def band_image(self) -> torch.Tensor:
        """(N, rows, S, C+1) view of the send buffer the forward kernel writes through (strided over cameras)"""
        return self.fx.image

    # -- forward -------------------------------------------------------------------------------------------------------
    def forward(self, world, normals, h, M, V, znear, zfar, first, num, feats, vr6=None, frame_n=None, order_refresh: int = 0,
                workspace_state=None, want_zbuf: bool = True):
        """[setup + binning] -> [fine + blend] of this rank's rows (``dss_render_forward`` with ``rows``); the RGBA band lands
        in the exchange's send buffer, the visibility flags in its flag buffer.  -> the dict of ``ops.render_forward``."""
        p = self.part
        if self._image_pending:
            # the asynchronous image exchange of the previous step reads the send buffer this forward overwrites: order the
            # compute stream behind it (long finished in a training loop: no stall)
            self.fx.finish()
            self._image_pending = False
        cutoff, thr, sigma, backface = self.settings
        kw = {"order_refresh": int(order_refresh)} if workspace_state is None else {"workspace_state": workspace_state}
        band_only = p.world_size > 1
        self.f = ops.render_forward(world, normals, h, M, V, znear, zfar, first, num, feats, self.S, self.K, cutoff, thr,
                                    sigma, backface, self.shared, rows=p.rows, out_image=self.fx.image,
                                    out_visible=self.fx.visible, vr6=vr6, frame_normals=frame_n, want_zbuf=want_zbuf,
                                    band_outputs_only=band_only,
                                    point_outputs=self._point_outputs if (band_only and self.static_buffers) else None, **kw)
        return self.f

    def start_exchange(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Issue the end-of-forward collectives: the visibility union (critical: the backward needs it) and -- unless
        `gather_image` is off or `late_image` defers it -- the image all-gather.  -> union of the flags, uint8 (P,)"""
        fx = self.fx
        if not self.active:
            self.vis_all = fx.visible
            return self.vis_all
        if fx.fold:
            self.vis_all = fx.start(out)
            self._image_pending = True
        elif self.late_image or not self.gather_image:
            self.vis_all = fx.start_visibility(out)
        else:
            self.vis_all = fx.start(out)
            self._image_pending = True
        return self.vis_all

    def start_image(self) -> None:
        """`late_image`: the image all-gather issued by the caller behind the backward's launches"""
        if self.active and self.gather_image and not self.fx.fold:
            self.fx.start_image()
            self._image_pending = True

    def full_image(self) -> torch.Tensor:
        """Wait for the image all-gather -> the full render (N, S, S, C+1) (a strided view of the receive buffer for equal
        contiguous bands, of the row-ordered copy otherwise)."""
        if not self.active:
            return self.fx.image
        if not self.gather_image:
            raise RuntimeError("this RowShardedRender was built with gather_image=False")
        self._image_pending = False
        return self.fx.finish()

    # -- backward ------------------------------------------------------------------------------------------------------
    # Five stages, compute and collectives alternating, so that a caller can replay the compute stages as captured graphs
    # around host-issued collectives (bench.py `graph_segments`) or time them apart; `backward` runs them in order.
    def _mark(self, label):
        if self.mark is not None:
            self.mark(label)

    def alpha_send_view(self):
        """(N, rows, S) view of the alpha-plane exchange's send buffer, or None when this step has no such exchange (bucket form,
        a world of one): a loss kernel that produces the band's gradient can write its alpha channel there directly
        (`ops.image_loss_band_backward_partials(alpha_out=...)`; then `bwd_begin(..., alpha_packed=True)`)"""
        if not (self.owner and self.part.world_size > 1):
            return None
        if self.alpha_x is None:
            self.alpha_x = AlphaPlaneExchange(self.part, self.N, self.dev, group=self.group)
        return self.alpha_x.send[:self.part.n_rows].permute(1, 0, 2)

    def bwd_begin(self, grad, alpha_packed: bool = False, full=None):
        """stage 1 (compute): the band's gradient, contiguous; band loss + owner form: the alpha channel packed for its
        exchange (unless the producer of ``grad`` already wrote it into `alpha_send_view()`).  ``grad``: gradient of the FULL
        image (N,S,S,C+1) -- replicated loss -- or of this rank's band."""
        p, S = self.part, self.S
        rows = p.n_rows
        banded = p.world_size > 1          # (a forced world of one owns every row: the plain backward, then the collectives)
        # whether `grad` is the full image's gradient decides which COLLECTIVES the step issues (band loss + owner form: the
        # alpha-plane exchange): every rank must decide alike, so the caller says so (`full`); inferring it from the shape is
        # only unambiguous when no rank owns all the rows or none
        if full is None:
            full = banded and grad.shape[1] == S and rows != S
        full = bool(full) and banded
        if grad.shape[1] != (S if full else rows):
            raise RuntimeError("grad must be the full image gradient (N,%d,S,C+1) or the band's (N,%d,S,C+1), got %s (full=%s)"
                               % (S, rows, tuple(grad.shape), full))
        st = {"full": full, "banded": banded, "grad": grad.contiguous() if full else None}
        st["g_band"] = p.slice(grad).contiguous() if full else grad.contiguous()
        st["alpha"] = self.owner and banded and not full
        if st["alpha"]:
            if self.alpha_x is None:
                self.alpha_x = AlphaPlaneExchange(self.part, self.N, self.dev, group=self.group)
            if not alpha_packed:
                self.alpha_x.pack(st["g_band"])
        self._bwd = st
        return st

    def bwd_exchange_alpha(self):
        """stage 2 (collective, band loss + owner form only): all-gather of the alpha channel of the bands' gradients"""
        if self._bwd["alpha"]:
            self.alpha_x.gather()

    def bwd_compute(self, radii_s: float, clip: float, world, M, V, first, num, f=None, vis_all=None):
        """stage 3 (compute): the fused backward of this rank's rows (+ owner form: clip + projection in front of the
        reduction)"""
        p, S, C, P, Pw = self.part, self.S, self.C, self.P, self.Pw
        st = self._bwd
        f = self.f if f is None else f
        vis_all = self.vis_all if vis_all is None else vis_all
        st["f"] = f
        extra = {}
        if self.owner and st["banded"]:
            if st["full"]:
                extra["grad_out_full"] = st["grad"]
            else:
                extra["grad_occ_full"] = self.alpha_x.plane_in_image_order()
        if self.owner:
            nf = Pw if self.features_shared else P
            g_world = self.wbucket[:Pw * 3].view(Pw, 3)
            g_fsum = self.wbucket[Pw * 3:].view(nf, C)
            # feature partials: straight into the reduction buffer unless they are summed over the cameras first
            g_feat = self.bucket[:P * C].view(P, C) if self.features_shared else g_fsum
            g_pts = self._pts_scratch
            ops.render_backward(st["g_band"], f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"],
                                vis_all, first, num, radii_s, -1.0, image_size=S, rows=p.rows, out=(g_feat, g_pts), **extra)
            self._mark("backward_compute")
            # every pair's position gradient is complete on its owner (zero elsewhere): clip + projection + the sum over
            # the cameras first, then ONE all-reduce of the world-space sums
            if self.features_shared:
                ops.project_backward(world, M, V, first, num, g_pts, f["valid"], self.shared, clip=clip, grad_features=g_feat,
                                     out=(g_world, g_fsum))
            else:
                ops.project_backward(world, M, V, first, num, g_pts, f["valid"], self.shared, clip=clip, out=(g_world, None))
            self._mark("projection_compute")
            st["out"] = (g_world, g_fsum)
            return
        g_feat = self.bucket[:P * C].view(P, C)
        g_pts = self.bucket[P * C:].view(P, 3)
        ops.render_backward(st["g_band"], f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], vis_all,
                            first, num, radii_s, -1.0, image_size=S, rows=p.rows, out=(g_feat, g_pts))
        self._mark("backward_compute")
        st["out"] = (g_feat, g_pts)

    def bwd_reduce(self):
        """stage 4 (collective): ONE all-reduce (SUM) -- owner: the world-space sums; bucket: the per-pair partial sums"""
        if self.active:
            dist.all_reduce(self.wbucket if self.owner else self.bucket, op=dist.ReduceOp.SUM, group=self.group)
        self._mark("wait_gradient_allreduce")

    def bwd_finish(self, clip: float, world, M, V, first, num):
        """stage 5 (compute, bucket form only): the per-point clip is non-linear, so it follows the sum over the ranks --
        inside the projection launch.  -> (grad_world (Pw,3), grad_features)"""
        st = self._bwd
        if self.owner:
            return st["out"]
        C, P, Pw = self.C, self.P, self.Pw
        g_feat, g_pts = st["out"]
        f = st["f"]
        g_world = self.wbucket[:Pw * 3].view(Pw, 3)
        if self.features_shared:
            g_fsum = self.wbucket[Pw * 3:].view(Pw, C)
            ops.project_backward(world, M, V, first, num, g_pts, f["valid"], self.shared, clip=clip, grad_features=g_feat,
                                 out=(g_world, g_fsum))
            self._mark("projection_compute")
            return g_world, g_fsum
        ops.project_backward(world, M, V, first, num, g_pts, f["valid"], self.shared, clip=clip, out=(g_world, None))
        self._mark("projection_compute")
        return g_world, g_feat

    def backward(self, grad, radii_s: float, clip: float, world, M, V, first, num, f=None, vis_all=None, full=None):
        """Backward of this rank's rows + the gradient exchange.  ``grad`` is either the gradient of the FULL image
        (N, S, S, C+1) -- replicated loss -- or of the rank's band (N, rows, S, C+1) -- band loss.
        -> (grad_world (Pw,3), grad_features ((Pw,C) if `features_shared` else (P,C))): the sums over all ranks, identical on
        every rank, in buffers this object owns (overwritten by the next backward).  ``full``: say which of the two it is
        (every rank must take the same branch; None infers it from the shape, ambiguous when a rank owns all rows or none)."""
        self.bwd_begin(grad, full=full)
        self.bwd_exchange_alpha()
        self._mark("wait_alpha_allgather")
        self.bwd_compute(radii_s, clip, world, M, V, first, num, f=f, vis_all=vis_all)
        if self.late_image:
            self.start_image()
        self.bwd_reduce()
        return self.bwd_finish(clip, world, M, V, first, num)
