"""Surface-splatting renderer: the MI355X drop-in for ``DSS.core.renderer`` (renderer.py:15-82).

``SurfaceSplattingRenderer(rasterizer, compositor, antialiasing_sigma=1.0, density=1e-4,
frnn_radius=-1).forward(point_clouds, **kwargs)`` returns the (N,H,W,4) RGBA image (alpha =
occupancy), or ``None`` for an empty cloud (:41-42), or ``(images, fragments)`` with ``verbose``.
Weights, NormWeightedCompositor and the RGBA concat (:53-78) run as ONE fused HIP kernel each way.
"""
import os

import torch
import torch.autograd as autograd

from . import ops

__all__ = ["SurfaceSplattingRenderer", "RowShardedSurfaceSplattingRenderer", "NormWeightedCompositor"]


class _Blend(autograd.Function):
    @staticmethod
    def forward(ctx, features, occupancy, idx, qvalue, scaler, geometry):
        out, wsum = ops.blend_forward(idx, qvalue, occupancy, scaler, features, return_wsum=True)
        ctx.save_for_backward(idx, qvalue, scaler, wsum)
        ctx.geometry = geometry
        ctx.num_points = features.shape[0]
        ctx.mark_non_differentiable(wsum)
        return out, wsum

    @staticmethod
    def backward(ctx, grad_out, _grad_wsum=None):
        idx, qvalue, scaler, wsum = ctx.saved_tensors
        grad_feat, grad_occ = ops.blend_backward(grad_out.contiguous(), idx, qvalue, scaler, ctx.num_points,
                                                 geometry=ctx.geometry, wsum=wsum)
        return grad_feat, grad_occ, None, None, None, None


class NormWeightedCompositor(torch.nn.Module):
    """Marker / stand-alone equivalent of pytorch3d.renderer.NormWeightedCompositor
    (``compositor_type`` in configs/default.yaml:31).  Called the pytorch3d way
    ``(idx (N,K,H,W) long, weights (N,K,H,W), features (C,P))`` it returns (N,C,H,W)."""

    def forward(self, fragments_idx, weights, features, **kwargs):
        idx = fragments_idx.permute(0, 2, 3, 1).to(torch.int32).contiguous()
        w = weights.permute(0, 2, 3, 1).contiguous()
        # weights are final: encode them as q = -2 ln w with unit scaler
        q = torch.where(idx >= 0, -2.0 * torch.log(w.clamp_min(1e-38)), torch.full_like(w, -1.0))
        feat = features.permute(1, 0).contiguous()
        occ = (idx[..., 0] >= 0).float()
        ones = torch.ones(feat.shape[0], device=feat.device)
        out, _ = _Blend.apply(feat, occ, idx, q, ones, None)
        return out[..., :-1].permute(0, 3, 1, 2)


class SurfaceSplattingRenderer(torch.nn.Module):
    def __init__(self, rasterizer, compositor=None, antialiasing_sigma: float = 1.0, density: float = 1e-4,
                 frnn_radius=-1, fused=None, graphed: bool = False, order_refresh: int = 0, engine_thread=None,
                 row_partition=None, gradient_exchange: str = "auto", row_output: str = "full", process_group=None):
        """``fused`` (not in the reference signature): True runs rasterizer + blend as ONE autograd node on the fused
        kernels (dss_render_forward / dss_render_backward): same images, ~2x fewer launches; the only loss of generality
        is that gradients w.r.t. ``fragments.zbuf`` are not propagated.  False keeps rasterizer and blend as separate
        autograd nodes.  None (default -- what `config.create_renderer` builds from the reference's YAML, which cannot
        name the argument): fused unless the call asks for the fragments (``verbose=True``), the only way a caller can
        put a loss on ``fragments.zbuf``.
        ``graphed`` (not in the reference signature either; implies the fused path): forward and backward replay as two
        hipGraphs over static buffers (`dss_amd.rasterizer._GraphedRender`): the host cost of an iteration drops to two graph
        launches.  The graphs read the point / normal / colour tensors in place, so keep handing over the SAME tensors
        (parameters updated in place); one render in flight per renderer: the returned image is overwritten by the
        next call.
        ``order_refresh`` = k > 0 (fused path, clouds above 2M points; not in the reference signature): the renderer keeps the
        screen-cell order its binning sorts the points into and reuses it for the next k - 1 renders of the same shape
        (`include/dss_hip.h` DSS_WS_ORDER_SAVE / DSS_WS_ORDER_REUSE): a training loop moves its points a little per iteration,
        so those renders skip the sort.  Images, fragments and gradients are identical bit for bit.
        ``engine_thread`` (not in the reference signature; default None = PyTorch's autograd state is left ALONE): an
        explicit ``engine_thread=False`` -- or the environment variable ``DSS_AMD_CALLING_THREAD_BACKWARD=1`` -- switches the
        CONSTRUCTING thread to backward-on-the-calling-thread (``torch.autograd.set_multithreading_enabled(False)``,
        thread-local, for everything that thread differentiates afterwards: a knowing, process-level opt-in).  The scoped form
        is ``with dss_amd.calling_thread_backward(): loss.backward()``, which restores the state on exit; see there for why
        it matters at DSS sizes (0.25 vs 0.125 ms of host time per forward + backward).  Round 4 flipped the switch in every
        constructor; a drop-in for the reference's renderer must not change global engine state as a side effect.
        ``row_partition`` (multi-GPU, not in the reference, which has no distributed layer): a
        `dss_amd.distributed.RowPartition`, or "auto" / "cyclic" / "bands" = this rank's share of the initialised
        ``torch.distributed`` process group (``process_group``, default the world; no group or a world of one: the plain
        single-GPU path).  Every rank then renders its image rows only and the call returns, through ONE autograd node
        (`dss_amd.rasterizer._RenderRowSharded`), with ``row_output="full"`` the whole (N,H,W,4) image gathered from all ranks
        -- an unmodified training loop evaluates its loss on it on every rank -- or with ``row_output="band"`` the rank's own
        rows (N,rows,W,4) for `dss_amd.distributed.band_image_loss`; ``loss.backward()`` leaves the SAME gradients (the sums
        over all ranks) on every rank.  ``gradient_exchange``: "owner", "bucket" or "auto" (default: chosen from the bytes each form
        puts on the critical path, `dss_amd.sharded.choose_gradient_exchange`)."""
        super().__init__()
        if engine_thread is None and os.environ.get("DSS_AMD_CALLING_THREAD_BACKWARD", "0") == "1":
            engine_thread = False
        self.engine_thread = engine_thread
        if engine_thread is False and torch.autograd.is_multithreading_enabled():
            torch.autograd.set_multithreading_enabled(False)
        self.fused = fused
        self.graphed = bool(graphed)
        self.order_refresh = int(order_refresh)
        if row_output not in ("full", "band"):
            raise ValueError("row_output must be 'full' or 'band', got %r" % (row_output,))
        self.row_partition, self.gradient_exchange, self.row_output = row_partition, gradient_exchange, row_output
        self.process_group = process_group
        self._auto_part = None
        self.rasterizer = rasterizer
        self.compositor = compositor
        self.cameras = self.rasterizer.cameras
        self._Vrk_h = None
        self.antialiasing_sigma = antialiasing_sigma
        self.density = density
        self.frnn_radius = frnn_radius

    def _is_norm_weighted(self) -> bool:
        # ours, or pytorch3d.renderer.NormWeightedCompositor (same semantics: recognised by name so that a
        # pytorch3d object configured in YAML takes the fused HIP path instead of pytorch3d's CUDA kernels)
        return isinstance(self.compositor, NormWeightedCompositor) or \
            type(self.compositor).__name__ == "NormWeightedCompositor"

    def _partition(self, raster_settings):
        """the row partition of this call: the object given, or this rank's share of the process group (resolved per image
        size, once), or None = single-GPU path"""
        rp = self.row_partition
        if rp is None:
            return None
        st = self.rasterizer.raster_settings if raster_settings is None else raster_settings
        S = int(st.image_size)
        if isinstance(rp, str):
            hit = self._auto_part
            if hit is None or hit[0] != S:
                from .sharded import default_partition
                hit = self._auto_part = (S, default_partition(S, self.process_group, layout=rp))
            return hit[1]
        if rp.S != S:
            raise ValueError("row_partition is for %d image rows, the raster settings say %d" % (rp.S, S))
        return rp if rp.world_size > 1 else None

    def forward(self, point_clouds, **kwargs):
        if point_clouds.isempty():
            return None
        fragments = kwargs.get("fragments", None)
        fused = (not kwargs.get("verbose", False)) if self.fused is None else bool(self.fused)
        part = self._partition(kwargs.get("raster_settings")) if fragments is None else None
        if part is not None:
            # multi-GPU: the fused path on this rank's rows, or nothing -- silently rendering the whole image on every rank
            # would hide a mis-configured job
            if not (hasattr(self.rasterizer, "render_fused") and self._is_norm_weighted()
                    and not self.rasterizer.compacts(kwargs.get("raster_settings"))):
                raise RuntimeError("a row-partitioned render needs dss_amd's SurfaceSplatting, a NormWeightedCompositor and "
                                   "the masked culling path (backface_culling off or compact_culled=False)")
            kw = {k: v for k, v in kwargs.items() if k != "fragments"}
            kw.update(row_partition=part, row_partition_auto=isinstance(self.row_partition, str) and self.row_partition == "auto",
                      gradient_exchange=self.gradient_exchange, process_group=self.process_group,
                      band_only=kwargs.get("band_only", self.row_output == "band"),
                      want_fragments=bool(kwargs.get("verbose", False)))
            if self.order_refresh > 0 and "order_refresh" not in kw:
                kw["order_refresh"] = self.order_refresh
            images, fragments, point_clouds = self.rasterizer.render_fused(point_clouds, **kw)
            if images.shape[-1] != 4:
                images = torch.cat([images[..., :3], images[..., -1:]], dim=-1)
            return (images, fragments) if kwargs.get("verbose", False) else images
        if (fragments is None and fused and hasattr(self.rasterizer, "render_fused")
                and not self.rasterizer.compacts(kwargs.get("raster_settings"))   # (that mode rebuilds the clouds first: unfused)
                and self._is_norm_weighted()
                and (point_clouds.features_packed() is None or point_clouds.features_packed().shape[1] <= 8)  # render_fused: C <= 8
                and self.rasterizer.raster_settings.points_per_pixel <= 32):
            kw = {k: v for k, v in kwargs.items() if k != "fragments"}
            kw["want_fragments"] = bool(kwargs.get("verbose", False))   # (only then are the fragment tensors materialised)
            if self.graphed:
                kw["graphed"] = True
            if self.order_refresh > 0 and "order_refresh" not in kw:
                kw["order_refresh"] = self.order_refresh
            images, fragments, point_clouds = self.rasterizer.render_fused(point_clouds, **kw)
            if images.shape[-1] != 4:  # RGBA contract of renderer.py:75-78: first three feature channels + occupancy
                images = torch.cat([images[..., :3], images[..., -1:]], dim=-1)
            return (images, fragments) if kwargs.get("verbose", False) else images
        if fragments is None:
            if kwargs.get("verbose", False):
                fragments, point_clouds, _ = self.rasterizer(point_clouds, **kwargs)
            else:
                fragments, point_clouds = self.rasterizer(point_clouds, **kwargs)
        pts_rgb = point_clouds.features_packed()[:, :3].contiguous()
        scaler = getattr(fragments, "scaler_packed", None)   # ours: per point (the gather is fused into the blend)
        if scaler is None:
            scaler = fragments.scaler
        if scaler.dim() != 1:
            # reference-style per-fragment scaler (N,H,W,K): fold it into q (w = exp(-q/2) * s)
            qv = torch.where(fragments.idx >= 0,
                             fragments.qvalue - 2.0 * torch.log(scaler.clamp_min(1e-38)), fragments.qvalue)
            scaler = torch.ones(pts_rgb.shape[0], device=pts_rgb.device)
        else:
            qv = fragments.qvalue
        if self.compositor is None or self._is_norm_weighted():
            images, wsum = _Blend.apply(pts_rgb, fragments.occupancy, fragments.idx, qv, scaler,
                                        getattr(fragments, "geometry", None))
            if self.compositor is None:
                # renderer.py:59-65: without a compositor the reference calls pytorch3d's `weighted_sum`, the
                # UN-normalised sum_k f_k w_k (weights get no gradient there either).  The kernel divides by
                # wsum = max(sum_k w_k, 1e-4); multiplying back by the same (constant) wsum undoes exactly that.
                images = torch.cat([images[..., :-1] * wsum.unsqueeze(-1), images[..., -1:]], dim=-1)
        else:
            # foreign compositor object: call it exactly like renderer.py:53-78
            safe = fragments.idx.clamp_min(0).long()
            w = torch.exp(-0.5 * qv) * scaler[safe] * (fragments.idx >= 0)
            images = self.compositor(fragments.idx.long().permute(0, 3, 1, 2), w.permute(0, 3, 1, 2),
                                     pts_rgb.permute(1, 0), **kwargs)
            images = torch.cat([images.permute(0, 2, 3, 1), fragments.occupancy.unsqueeze(-1)], dim=-1)
        if kwargs.get("verbose", False):
            return images, fragments
        return images


class RowShardedSurfaceSplattingRenderer(SurfaceSplattingRenderer):
    """`SurfaceSplattingRenderer` that shards its image rows over the GPUs of the node -- the class a YAML names to run the
    reference's unmodified training script on several GPUs (``config.py:241-261`` can pass no constructor argument, only a
    class path)::

        renderer:
          renderer_type: dss_amd.renderer.RowShardedSurfaceSplattingRenderer
          raster_type: dss_amd.rasterizer.SurfaceSplatting
          compositor_type: dss_amd.renderer.NormWeightedCompositor

        torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_mvr.py --config cfg.yml

    Every rank runs the same script on the same batches (the reference seeds its RNGs identically in every process,
    ``DSS/__init__.py:12-16``): a replicated loop around a sharded render.  The constructor is where the process joins the job
    -- an explicit opt-in by class choice: when ``WORLD_SIZE`` > 1 is in the environment and no process group exists yet it
    selects the GPU ``LOCAL_RANK`` (``train_mvr.py:32-33`` then places the model on the current device) and initialises
    ``torch.distributed`` (backend ``nccl`` = RCCL; ``DSS_AMD_DIST_BACKEND`` overrides, e.g. ``gloo`` in tests)."""

    def __init__(self, rasterizer, compositor=None, **kwargs):
        import torch.distributed as dist
        kwargs.setdefault("row_partition", os.environ.get("DSS_AMD_ROW_PARTITION", "auto"))
        kwargs.setdefault("gradient_exchange", os.environ.get("DSS_AMD_GRADIENT_EXCHANGE", "auto"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and dist.is_available() and not dist.is_initialized():
            if torch.cuda.is_available():
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("DSS_AMD_DIST_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
            else:
                dist.init_process_group(backend)
        super().__init__(rasterizer, compositor, **kwargs)
