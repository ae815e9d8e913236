"""Surface-splatting rasterizer: the MI355X drop-in for ``DSS.core.rasterizer``.

Same public names, constructor / ``forward`` signatures and return tuples as the reference
(DSS/core/rasterizer.py): ``PointFragments`` (:31-36), ``PointsRasterizationSettings`` (:39-99),
``SurfaceSplatting`` (:102-664), ``rasterize_elliptical_points`` (:681-744) and the
``EllipticalRasterizer`` autograd.Function (:747-977).  ``config.py:241-261`` resolves
``raster_type: dss_amd.rasterizer.SurfaceSplatting`` and looks ``PointsRasterizationSettings`` up in
the same module, so both are exported here.  All device work goes through the C ABI
(``dss_amd.ops``); there is no torch/CPU fallback.

Differences that a caller can observe (documented in DESIGN.md / INTEGRATION.md):
* culled points are masked, not compacted: the returned point cloud is the cloud extended to the
  N cameras, ``fragments.idx`` indexes its packed points, culled points simply never appear
  (``SurfaceSplatting(..., compact_culled=True)`` drops them like the reference: same integer labels, h from the
  filtered clouds -- at the price of boolean indexing with host syncs per call; by default that mode is taken exactly
  when ``raster_settings.backface_culling`` is on, where masking would change h and therefore the image);
* a point exactly on a pixel centre contributes 0 to the occupancy gradient (reference: NaN).
"""
from typing import Optional

import torch
import torch.autograd as autograd

from . import neighbours, ops
from .cloud import PointClouds3D, shared_cloud_ranges

__all__ = ["PointFragments", "PointsRasterizationSettings", "SurfaceSplatting", "rasterize_elliptical_points",
           "EllipticalRasterizer", "knn_variance_scale"]


class PointFragments(tuple):
    """rasterizer.py:31-36: ``(idx, zbuf, qvalue, scaler, occupancy)`` with the reference's shapes -- a real ``tuple``
    subclass like the reference's NamedTuple (``isinstance(fragments, tuple)``, unpacking, indexing, ``len``, ``_fields``,
    ``_replace`` / ``_asdict``), with named fields.

    ``scaler`` is the per-FRAGMENT ``(N, H, W, K)`` tensor of rasterizer.py:631-633 (0 where ``idx < 0``), but it is only
    materialised when somebody reads it (field, index 3, iteration): the kernels consume the per-point ``scaler_packed
    (P,)`` (the gather is fused into the blend).  ``geometry`` = (pts_screen, radii, visible, first_idx, num_points) lets
    the blend backward run as a deterministic point-centric gather instead of an atomic scatter; None is always legal."""
    _fields = ("idx", "zbuf", "qvalue", "scaler", "occupancy")

    def __new__(cls, idx, zbuf, qvalue, scaler, occupancy, geometry=None):
        packed = scaler is not None and scaler.dim() == 1   # per point: keep packed, gather lazily
        self = tuple.__new__(cls, (idx, zbuf, qvalue, None if packed else scaler, occupancy))
        self.geometry = geometry
        self.scaler_packed = scaler if packed else None
        self._scaler = None if packed else scaler
        return self

    idx = property(lambda self: tuple.__getitem__(self, 0))
    zbuf = property(lambda self: tuple.__getitem__(self, 1))
    qvalue = property(lambda self: tuple.__getitem__(self, 2))
    occupancy = property(lambda self: tuple.__getitem__(self, 4))

    @property
    def scaler(self):
        if self._scaler is None and self.scaler_packed is not None:
            idx = self.idx
            self._scaler = torch.where(idx >= 0, self.scaler_packed[idx.clamp_min(0).long()],
                                       torch.zeros((), dtype=self.scaler_packed.dtype, device=idx.device))
        return self._scaler

    def __iter__(self):
        return iter((self.idx, self.zbuf, self.qvalue, self.scaler, self.occupancy))

    def __getitem__(self, i):
        return (self.idx, self.zbuf, self.qvalue, self.scaler, self.occupancy)[i]

    def __eq__(self, other):
        return self is other

    def __ne__(self, other):
        return self is not other

    __hash__ = object.__hash__

    def __reduce__(self):   # (pickling / copy: the materialised form)
        return (PointFragments, (self.idx, self.zbuf, self.qvalue,
                                 self.scaler_packed if self.scaler_packed is not None else self._scaler, self.occupancy,
                                 self.geometry))

    def _asdict(self):
        return dict(zip(self._fields, tuple(iter(self))))

    def _replace(self, **kw):
        d = dict(idx=self.idx, zbuf=self.zbuf, qvalue=self.qvalue, occupancy=self.occupancy, geometry=self.geometry,
                 scaler=self.scaler_packed if self.scaler_packed is not None else self._scaler)
        d.update(kw)
        return PointFragments(**d)

    def __repr__(self):
        return "PointFragments(idx=%r, zbuf=%r, qvalue=%r, scaler=<%s>, occupancy=%r)" % (
            tuple(self.idx.shape), None if self.zbuf is None else tuple(self.zbuf.shape), tuple(self.qvalue.shape),
            "packed (P,)" if self._scaler is None else "per fragment", tuple(self.occupancy.shape))


class PointsRasterizationSettings:
    """rasterizer.py:39-99 (same slots and defaults)."""
    __slots__ = ["cutoff_threshold", "backface_culling", "depth_merging_threshold", "Vrk_invariant",
                 "Vrk_isotropic", "radii_backward_scaler", "image_size", "points_per_pixel", "bin_size",
                 "max_points_per_bin", "clip_pts_grad", "antialiasing_sigma"]

    def __init__(self, backface_culling: bool = True, cutoff_threshold: float = 1,
                 depth_merging_threshold: float = 0.05, Vrk_invariant: bool = False, Vrk_isotropic: bool = True,
                 radii_backward_scaler: float = 10, image_size: int = 256, points_per_pixel: int = 8,
                 bin_size: Optional[int] = 0, max_points_per_bin: Optional[int] = None,
                 clip_pts_grad: Optional[float] = -1, antialiasing_sigma: Optional[float] = 1.0):
        self.cutoff_threshold = cutoff_threshold
        self.backface_culling = backface_culling
        self.depth_merging_threshold = depth_merging_threshold
        self.Vrk_invariant = Vrk_invariant
        self.Vrk_isotropic = Vrk_isotropic
        self.radii_backward_scaler = radii_backward_scaler
        self.image_size = image_size
        self.points_per_pixel = points_per_pixel
        self.bin_size = bin_size
        self.max_points_per_bin = max_points_per_bin
        self.clip_pts_grad = clip_pts_grad
        self.antialiasing_sigma = antialiasing_sigma


def _raster_forward(ctx, pts_screen, ellipse_param, cutoff_threshold, radii, cloud_to_packed_first_idx,
                    num_points_per_cloud, depth_merging_threshold, image_size, points_per_pixel, bin_size,
                    max_points_per_bin, radii_backward_scaler, clip_pts_grad):
    idx, zbuf, qvalue_map, occ_map, visible = ops.splat_points(
        pts_screen, ellipse_param, cutoff_threshold, radii, cloud_to_packed_first_idx, num_points_per_cloud,
        depth_merging_threshold, image_size, points_per_pixel, bin_size, max_points_per_bin, return_visible=True)
    ctx.radii_backward_scaler = radii_backward_scaler
    ctx.clip_pts_grad = -1.0 if clip_pts_grad is None else float(clip_pts_grad)
    ctx.save_for_backward(pts_screen, radii, idx, visible, cloud_to_packed_first_idx, num_points_per_cloud)
    # unused output gradients arrive as None instead of dense zero tensors: an all-zero zbuf gradient (train_mvr.py:
    # zbuf feeds no loss) then costs neither an (N,S,S,K) allocation nor the scatter pass + separate clip launch
    ctx.set_materialize_grads(False)
    return idx, zbuf, qvalue_map, occ_map, visible


def _raster_backward(ctx, zbuf_grad, occ_grad):
    # qvalue_grad is ignored exactly like the reference (rasterizer.py:788-789)
    pts_screen, radii, idx, visible, first_idx, num_points = ctx.saved_tensors
    if occ_grad is None and zbuf_grad is None:
        return torch.zeros_like(pts_screen)
    return ops.splat_backward(pts_screen, radii, visible, idx, occ_grad, zbuf_grad, first_idx, num_points,
                              ctx.radii_backward_scaler, ctx.clip_pts_grad)


class EllipticalRasterizer(autograd.Function):
    """rasterizer.py:747-977.  ``apply(pts_screen, ellipse_param, cutoff_threshold, radii,
    cloud_to_packed_first_idx, num_points_per_cloud, depth_merging_threshold, image_size,
    points_per_pixel, bin_size, max_points_per_bin, radii_backward_scaler[, clip_pts_grad])``
    -> ``(idx, zbuf, qvalue_map, occ_map)``; the backward produces a gradient for ``pts_screen`` only
    (rasterizer.py:975-977): occupancy surrogate on xy + zbuf scatter on z."""

    @staticmethod
    def forward(ctx, pts_screen, ellipse_param, cutoff_threshold, radii, cloud_to_packed_first_idx,
                num_points_per_cloud, depth_merging_threshold, image_size, points_per_pixel, bin_size=0,
                max_points_per_bin=0, radii_backward_scaler=10.0, clip_pts_grad=-1.0):
        outs = _raster_forward(ctx, pts_screen, ellipse_param, cutoff_threshold, radii, cloud_to_packed_first_idx,
                               num_points_per_cloud, depth_merging_threshold, image_size, points_per_pixel, bin_size,
                               max_points_per_bin, radii_backward_scaler, clip_pts_grad)
        ctx.mark_non_differentiable(outs[0])
        return outs[:4]

    @staticmethod
    def backward(ctx, idx_grad, zbuf_grad, qvalue_grad, occ_grad):
        return (_raster_backward(ctx, zbuf_grad, occ_grad),) + (None,) * 12


class _EllipticalRasterizerWithVisibility(autograd.Function):
    """`EllipticalRasterizer` with the per-point visibility flags of the fine pass (the reference recomputes them with
    ``idx[mask].unique()``, utils/__init__.py:320-340) as a fifth, non-differentiable OUTPUT: what `SurfaceSplatting`
    calls, so that nothing travels through module or class state."""

    @staticmethod
    def forward(ctx, *args):
        outs = _raster_forward(ctx, *args)
        ctx.mark_non_differentiable(outs[0], outs[4])
        return outs

    @staticmethod
    def backward(ctx, idx_grad, zbuf_grad, qvalue_grad, occ_grad, visible_grad):
        return (_raster_backward(ctx, zbuf_grad, occ_grad),) + (None,) * 12


def rasterize_elliptical_points(pcls_screen, ellipse_params, cutoff_threshold, radii,
                                depth_merging_threshold: float = 0.05, image_size: int = 512,
                                points_per_pixel: int = 5, bin_size: Optional[int] = None,
                                max_points_per_bin: Optional[int] = None, radii_backward_scaler: float = 10.0,
                                clip_pts_grad: float = -1.0):
    """rasterizer.py:681-744.  ``pcls_screen`` is a point-cloud object whose packed points are
    (NDC x, NDC y, view z).  The per-point gradient clip hook (:735-737) is fused into the backward."""
    points_packed = pcls_screen.points_packed()
    cutoff_threshold = cutoff_threshold.expand(points_packed.shape[0])
    return EllipticalRasterizer.apply(points_packed, ellipse_params, cutoff_threshold, radii,
                                      pcls_screen.cloud_to_packed_first_idx(), pcls_screen.num_points_per_cloud(),
                                      depth_merging_threshold, image_size, points_per_pixel, bin_size,
                                      max_points_per_bin, radii_backward_scaler, clip_pts_grad)


class _ProjectAndSetup(autograd.Function):
    """Fused filter_renderable + transform + _get_per_point_info (rasterizer.py:219-254, 614, 525-565).
    Differentiable output: pts_screen (the EWA terms are detached in the reference, :562-565)."""

    @staticmethod
    def forward(ctx, world, normals, h, M, V, znear, zfar, first_idx, num_points, image_size, cutoff, sigma,
                backface, shared, vr6=None, frame_normals=None):
        info = ops.point_setup(world, normals, h, M, V, znear, zfar, first_idx, num_points, image_size, cutoff,
                               sigma, backface, shared, vr6=vr6, frame_normals=frame_normals)
        ctx.save_for_backward(world, M, V, first_idx, num_points, info["valid"])
        ctx.shared = shared
        outs = (info["pts_screen"], info["ellipse_params"], info["radii"], info["scaler"],
                info["cutoff_threshold"], info["valid"])
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_screen, *unused):
        world, M, V, first_idx, num_points, valid = ctx.saved_tensors
        gw = ops.project_backward(world, M, V, first_idx, num_points, g_screen.contiguous(), valid, ctx.shared)
        return (gw,) + (None,) * 15


_CAMERA_FIELDS = ("R", "T", "znear", "zfar", "fov", "aspect_ratio")


def _camera_state(cameras):
    """(name, object, version) of every camera field the projection depends on -- the key `_prepare` memoises on.  A field
    re-assigned to a fresh tensor changes identity, one modified in place changes version (a fresh tensor's version is 0
    again, so the version alone is no key; the tuple keeps the objects alive, so an id cannot be recycled either)."""
    return tuple((n, getattr(cameras, n, None), getattr(getattr(cameras, n, None), "_version", 0)) for n in _CAMERA_FIELDS)


def _kw_state(v):
    """memo key of a znear / zfar keyword: identity + version of a tensor, the value of a number, None"""
    if v is None:
        return None
    if torch.is_tensor(v):
        return (id(v), v._version)
    try:
        return ("v", float(v))
    except (TypeError, ValueError):
        return ("id", id(v))


def knn_variance_scale(point_clouds, K: int = 7) -> torch.Tensor:
    """Per-point 0.5 * (K-th smallest squared distance to the own cloud, self included), packed (P,)
    = ``0.5 * knn(...)[:, :, 1:].max(-1)`` of rasterizer.py:310-321 / 366-383, computed by the HIP grid
    search ``dss_knn_kth_sqdist`` (the reference uses FRNN / pytorch3d CUDA here)."""
    with torch.no_grad():
        d = ops.knn_kth_sqdist(point_clouds.points_packed().detach(), point_clouds.cloud_to_packed_first_idx(),
                               point_clouds.num_points_per_cloud(), K)
    return 0.5 * d


class SurfaceSplatting(torch.nn.Module):
    """rasterizer.py:102-664.  ``forward(point_clouds, point_clouds_filter=None, **kwargs)`` returns the
    tuple ``(PointFragments, point_clouds[, per_point_info])`` (:655-664)."""

    def __init__(self, cameras=None, raster_settings=None, frnn_radius=0.2, compact_culled: Optional[bool] = None,
                 detect_identical_clouds: bool = True):
        super().__init__()
        # N equal-sized clouds arriving as separate tensors are compared once per forward (one launch, one host read): if
        # they hold the same positions the neighbour statistic is searched in ONE of them (see _variance_scale)
        self.detect_identical_clouds = bool(detect_identical_clouds)
        if raster_settings is None:
            raster_settings = PointsRasterizationSettings()
        self.cameras = cameras
        self.raster_settings = raster_settings
        self.frnn_radius = frnn_radius
        # True: drop culled points like the reference (see _forward_compacted); False: mask them (no host syncs);
        # None (default): follow `raster_settings.backface_culling` -- with back-face culling on, roughly half of the
        # cloud is dropped and the reference's h (hence every splat size, hence the image) comes from the filtered
        # clouds, so a default-constructed PointsRasterizationSettings() (backface_culling=True like the reference's)
        # must take the reference's order to give the reference's image; the shipped configs (backface_culling: false)
        # keep the sync-free masked path.
        self.compact_culled = compact_culled
        self._Vrk_h = None

    def compacts(self, raster_settings=None) -> bool:
        """whether a forward with these settings takes the reference's drop-the-culled-points order"""
        if self.compact_culled is not None:
            return bool(self.compact_culled)
        rs = self.raster_settings if raster_settings is None else raster_settings
        return bool(getattr(rs, "backface_culling", False))

    # -- source-space variance scale h (rasterizer.py:293-402) ------------------------------------
    def _variance_scale(self, point_clouds, raster_settings, refresh=True, view=None):
        """``view`` = (V (N,4,4), znear (N,), zfar (N,), shared): the cameras of this render.  The reference computes the
        statistic AFTER `filter_renderable` has extended the cloud to the cameras and dropped, per camera, the points outside
        its depth range (rasterizer.py:599, 236-240, 183-217): the neighbours of a point are the ones the SAME camera keeps,
        Vrk_invariant takes one h per camera as a mean over the padded length of the batch (:325), Vrk_isotropic one h per
        (camera, point).  With ``view`` the search runs in that order (`ops.knn_kth_sqdist_view`: one grid build, one query
        launch with a grid row per camera); without it (callers outside a render) over the whole clouds.
        ``frnn_radius`` > 0 (the reference's default 0.2) selects its fixed-radius search, which reports neighbours beyond the
        radius as -1 (rasterizer.py:316-319): honoured -- it decides h once points have drifted away from the surface."""
        n_total = sum(p.shape[0] for p in point_clouds.points_list())
        if not refresh and self._Vrk_h is not None and (raster_settings.Vrk_invariant or
                                                       self._Vrk_h.shape[0] == n_total):  # rasterizer.py:359-361
            return self._Vrk_h
        first, num = point_clouds.cloud_to_packed_first_idx(), point_clouds.num_points_per_cloud()
        sizes = [p.shape[0] for p in point_clouds.points_list()]
        radius = self.frnn_radius if (self.frnn_radius is not None and self.frnn_radius > 0) else -1.0
        invariant, isotropic = bool(raster_settings.Vrk_invariant), bool(raster_settings.Vrk_isotropic)
        if not invariant and not isotropic:
            h = torch.zeros(n_total, device=point_clouds.device)   # unused: the anisotropic variance comes from _local_frames
            self._Vrk_h = h
            return h
        pts = point_clouds.points_packed().detach()
        if view is not None:
            V, znear, zfar, shared = view
            N = V.shape[0]
            if not shared and N > 1 and len(sizes) == N and min(sizes) == max(sizes) and sizes[0] > 0 and self.detect_identical_clouds:
                # N clouds of one size handed over as separate tensors: the reference's own texture does that to ONE model cloud
                # (`pointclouds.extend(N)` CLONES it N times, texture.py:88, before the colours are made per camera), and
                # its train_mvr.py reaches this rasterizer that way -- 8 x 99,790 points searched as 798k.  If the copies hold
                # the same positions, the neighbour statistic of every camera is that of the one cloud under the camera's
                # own culling: search the first copy as a cloud shared by the N cameras (same values, same (camera, point)
                # layout).  Costs one comparison launch and one host read; the render itself still takes the N clouds as
                # given, so the gradients keep their N separate ways back.
                with torch.no_grad():
                    copies = pts.view(N, sizes[0], 3)
                    if bool((copies[1:] == copies[:1]).all()):
                        shared, pts, first, num, sizes = True, copies[0], first[:1], num[:1], sizes[:1]
            with torch.no_grad():
                d = ops.knn_kth_sqdist_view(pts, first, num, 7, V, znear, zfar, shared, radius=radius)   # (N,Pw) shared, else (P,)
                if invariant:
                    f1, n1 = (first.new_zeros(N), num[:1].expand(N).contiguous()) if shared else (first, num)
                    h = ops.renderable_mean_clamp(d, pts, V, znear, zfar, f1, n1, shared, 0.5, 5e-5, 1e-3, 0.5e-3, 7)
                else:
                    # per point (rasterizer.py:383-388); for a shared cloud one value per (camera, point) pair, packed like
                    # the extended cloud
                    h = (0.5 * d).clamp_(5e-5, 0.01).reshape(-1)
                    if min(sizes) < 7:   # "knn search is unreliable, set sq_dist manually" (rasterizer.py:378-379)
                        small = torch.cat([torch.full((n,), n < 7, dtype=torch.bool) for n in sizes]).to(h.device)
                        small = small.repeat(N) if (shared and N > 1) else small
                        h = torch.where(small, torch.full_like(h, 0.5e-3), h)
            self._Vrk_h = h
            return h
        with torch.no_grad():
            # through dss_amd.neighbours: one search serves this statistic and the regularisers of the same iteration
            d = neighbours.kth_sqdist(pts, first, num, sizes, 7, radius=radius)
        if invariant:
            # one scalar per cloud: mean_i(0.5 max kNN-7 d^2) clamped to [5e-5, 1e-3]; clouds with fewer than
            # 7 points use sq_dist = 1e-3 (rasterizer.py:320-326)
            h = ops.cloud_mean_clamp(d, first, num, 0.5, 5e-5, 1e-3, 0.5e-3, 7)
        else:
            h = (0.5 * d).clamp_(5e-5, 0.01)  # per point (rasterizer.py:383-388)
            if min(sizes) < 7:  # "knn search is unreliable, set sq_dist manually" (rasterizer.py:378-379): 0.5 * 1e-3
                small = torch.cat([torch.full((n,), n < 7, dtype=torch.bool) for n in sizes]).to(h.device)
                h = torch.where(small, torch.full_like(h, 0.5e-3), h)
        self._Vrk_h = h
        return h

    def _local_frames(self, point_clouds):
        """Anisotropic source variance (rasterizer.py:256-291): PCA frames of the 8-neighbourhoods
        (estimate_pointcloud_local_coord_frames(neighborhood_size=8)) -> (vr6 (P,6), frame normals (P,3))."""
        first, num = point_clouds.cloud_to_packed_first_idx(), point_clouds.num_points_per_cloud()
        with torch.no_grad():
            pts = point_clouds.points_packed().detach()
            _, idx = ops.knn_points(pts, first, num, 8)
            return ops.local_frames(pts, idx, first, num)

    def _empty_fragments(self, batch_size, device, raster_settings):  # rasterizer.py:567-582
        S, K = raster_settings.image_size, raster_settings.points_per_pixel
        return PointFragments(idx=torch.full((batch_size, S, S, K), -1, dtype=torch.int32, device=device),
                              zbuf=torch.full((batch_size, S, S, K), -1.0, device=device),
                              qvalue=torch.full((batch_size, S, S, K), -1.0, device=device),
                              scaler=torch.zeros((batch_size, S, S, K), device=device),
                              occupancy=torch.zeros((batch_size, S, S), device=device))

    def _prepare(self, point_clouds, **kwargs):
        """Everything the kernels need from the (camera, cloud) objects: packed world points / normals,
        variance scale, camera matrices, cloud ranges.  One cloud is shared by all N cameras
        (Pointclouds.extend, rasterizer.py:236-240) or there are N clouds.

        A training loop calls this every iteration with the same tensors in a fresh cloud object: everything that
        only depends on WHICH tensors / cameras / settings these are (not on their values) is memoised on their
        identities (+ the cameras' version counters); a caller-supplied ``Vrk_h`` then makes the call free of launches."""
        raster_settings = kwargs.get("raster_settings", self.raster_settings)
        cameras = kwargs.get("cameras", self.cameras)
        if cameras is None:
            raise ValueError("Cameras must be specified either at initialization or in the forward pass")
        self.cameras = cameras
        h_given = kwargs.get("Vrk_h", None)
        memo_key = None
        pl, nl = point_clouds.points_list(), point_clouds.normals_list()
        # (only when the packed geometry IS the caller's tensor -- one cloud, or one tensor extended to the cameras: a
        # concatenation of several tensors is a copy that has to be rebuilt from their current values)
        if h_given is not None and (raster_settings.Vrk_invariant or raster_settings.Vrk_isotropic) and len(pl) >= 1 \
                and all(t is pl[0] for t in pl) and nl is not None and all(t is nl[0] for t in nl):
            cam_state = tuple(getattr(cameras, k, None) for k in ("R", "T", "znear", "zfar", "fov", "aspect_ratio"))
            memo_key = (id(cameras), tuple(id(t) for t in cam_state), tuple(getattr(t, "_version", 0) for t in cam_state),
                        tuple(id(t) for t in pl), None if nl is None else tuple(id(t) for t in nl),
                        tuple(t.shape[0] for t in pl), id(h_given), h_given._version, id(raster_settings),
                        _kw_state(kwargs.get("znear", None)), _kw_state(kwargs.get("zfar", None)))
            hit = getattr(self, "_prepare_memo", None)
            if hit is not None and hit[0] == memo_key:
                a = dict(hit[1])
                N = a["N"]
                a["out_clouds"] = point_clouds if (not a["shared"] or len(point_clouds) == N) else point_clouds.extend(N)
                a["raster_settings"] = raster_settings
                return a
        N = cameras.R.shape[0]
        dev = point_clouds.device
        shared = len(point_clouds) == 1 and N >= 1
        if not shared and len(point_clouds) != N:
            raise ValueError("need 1 or %d point clouds for %d cameras, got %d" % (N, N, len(point_clouds)))
        # `Pointclouds.extend(N)` (what the texture and the reference renderer hand over) is N references to the SAME
        # position / normal tensors, only the colours differ per camera: keep one copy of the geometry (shared-cloud
        # kernels, one kNN instead of N) and the per-camera colours
        geometry = point_clouds
        if not shared and N > 1:
            pl, nl = point_clouds.points_list(), point_clouds.normals_list()
            if all(t is pl[0] for t in pl) and nl is not None and all(t is nl[0] for t in nl):
                shared = True
                geometry = type(point_clouds)([pl[0]], [nl[0]]) if isinstance(point_clouds, PointClouds3D) else None
                if geometry is None:
                    shared, geometry = False, point_clouds
        M = cameras.get_full_projection_transform().get_matrix().to(dev, torch.float32).contiguous()
        V = cameras.get_world_to_view_transform().get_matrix().to(dev, torch.float32).contiguous()

        def as_n(v, d):
            t = getattr(cameras, v, kwargs.get(v, d))
            if (torch.is_tensor(t) and t.dtype == torch.float32 and t.device == dev and t.dim() == 1
                    and t.shape[0] == N and t.is_contiguous()):
                return t  # the usual case: no copy, no launch
            return torch.as_tensor(t, dtype=torch.float32, device=dev).reshape(-1).expand(N).contiguous()
        znear, zfar = as_n("znear", 1.0), as_n("zfar", 100.0)
        h = kwargs.get("Vrk_h", None)
        if h is None:
            h = self._variance_scale(geometry, raster_settings, kwargs.get("refresh", True), view=(V, znear, zfar, shared))
        vr6 = frame_n = None
        if not raster_settings.Vrk_invariant and not raster_settings.Vrk_isotropic:
            vr6, frame_n = self._local_frames(geometry)
        world, normals = geometry.points_packed(), geometry.normals_packed()
        if shared:
            Pc = world.shape[0]
            first_idx, num_points = shared_cloud_ranges(N, Pc, dev)  # cached: constant across iterations
            if h.numel() == 1:
                h = h.reshape(1).expand(N).contiguous()
            out_clouds = point_clouds if len(point_clouds) == N else point_clouds.extend(N)
        else:
            first_idx, num_points = point_clouds.cloud_to_packed_first_idx(), point_clouds.num_points_per_cloud()
            out_clouds = point_clouds
        a = dict(N=N, shared=shared, world=world, normals=normals, h=h.to(dev, torch.float32), M=M, V=V,
                 znear=znear, zfar=zfar, first_idx=first_idx, num_points=num_points,
                 out_clouds=out_clouds, raster_settings=raster_settings, vr6=vr6, frame_n=frame_n)
        if memo_key is not None and world is pl[0] and normals is nl[0]:
            # (only when the packed tensors ARE the caller's tensors: a duck-typed cloud whose points_packed() is a copy --
            # pytorch3d's torch.cat -- would otherwise be served a stale copy after an in-place optimiser step)
            # (the memo keeps the keyed objects alive -- an id cannot be recycled while it is the current entry)
            keep = (cameras, cam_state, pl, nl, h_given, raster_settings, kwargs.get("znear", None), kwargs.get("zfar", None))
            self._prepare_memo = (memo_key, {k: v for k, v in a.items() if k not in ("out_clouds", "raster_settings")}, keep)
        return a

    @staticmethod
    def _apply_activation_filter(point_clouds, point_clouds_filter):
        """rasterizer.py:230-234: a PointCloudsFilters object first drops the points whose ``activation`` flag is
        off (duck-typed: DSS.core.cloud.PointCloudsFilters.filter_with)."""
        if point_clouds_filter is not None and hasattr(point_clouds_filter, "filter_with"):
            return point_clouds_filter.filter_with(point_clouds, ("activation",))
        return point_clouds

    @staticmethod
    def _store_visibility(point_clouds_filter, visible, N, shared, original_clouds):
        """rasterizer.py:639-652: the per-point visibility of this render goes back into the filter object as a padded
        (N, P_max) mask over the ORIGINAL clouds -- points the activation filter dropped are never visible."""
        if point_clouds_filter is None or not hasattr(point_clouds_filter, "set_filter"):
            return
        vis = visible.view(torch.bool) if visible.dtype == torch.uint8 else visible.bool()   # 0 / 1 flags: no copy
        act = getattr(point_clouds_filter, "activation", None)
        num = original_clouds.num_points_per_cloud()
        sizes = [p.shape[0] for p in original_clouds.points_list()]
        p_max = int(max(sizes))
        dropped = (torch.is_tensor(act) and act.dim() == 2 and act.shape[1] > 1
                   and not (hasattr(point_clouds_filter, "all_on") and point_clouds_filter.all_on(act)))
        if not dropped and min(sizes) == p_max:       # the usual case: nothing to scatter, no host sync
            point_clouds_filter.set_filter(visibility=vis.view(N, p_max))
            return
        # rows = cameras (= rendered clouds); the packed flags are in row-major order of the kept positions
        keep = (torch.arange(p_max, device=vis.device)[None, :] < num.to(vis.device)[:, None]).expand(N, -1)
        if dropped:
            keep = keep & act.to(vis.device).bool()[:, :p_max].expand(N, -1)
        full = torch.zeros((N, p_max), dtype=torch.bool, device=vis.device)
        full.masked_scatter_(keep, vis)   # (`full[keep] = vis` counts the kept positions on the host: a sync per render)
        point_clouds_filter.set_filter(visibility=full)

    def _forward_compacted(self, point_clouds, original_clouds, point_clouds_filter, **kwargs):
        """``compact_culled=True``: the reference's exact order (rasterizer.py:219-254, 293-402): extend the cloud to the
        N cameras, DROP the points outside [znear, zfar] (and back faces) -- new, smaller clouds --, compute the variance
        scale h on those, rasterize them.  ``fragments.idx`` then labels the filtered packed cloud that is returned, integer
        for integer like the reference.  Costs boolean indexing with host syncs per call, which the default (masked)
        path avoids; results for every consumer of the (fragments, point_clouds) pair are the same either way."""
        raster_settings = kwargs.get("raster_settings", self.raster_settings)
        cameras = kwargs.get("cameras", self.cameras)
        if cameras is None:
            raise ValueError("Cameras must be specified either at initialization or in the forward pass")
        N = cameras.R.shape[0]
        dev = point_clouds.device
        pc = point_clouds if len(point_clouds) == N else point_clouds.extend(N)
        V = cameras.get_world_to_view_transform().get_matrix().to(dev, torch.float32)

        def per_cam(name, default):
            t = getattr(cameras, name, kwargs.get(name, default))
            return torch.as_tensor(t, dtype=torch.float32, device=dev).reshape(-1).expand(N)
        znear, zfar = per_cam("znear", 1.0), per_cam("zfar", 100.0)
        pl, nl, fl = pc.points_list(), pc.normals_list(), pc.features_list()
        keeps, pts_f, nrm_f, feat_f = [], [], [], []
        for n in range(N):
            with torch.no_grad():
                zview = pl[n] @ V[n, :3, 2] + V[n, 3, 2]
                keep = (zview >= znear[n]) & (zview <= zfar[n])
                if raster_settings.backface_culling:
                    keep &= (nl[n] @ V[n, :3, 2]) < 0
            keeps.append(keep)
            pts_f.append(pl[n][keep])
            nrm_f.append(nl[n][keep])
            if fl is not None:
                feat_f.append(fl[n][keep])
        try:
            filtered = type(point_clouds)(pts_f, nrm_f, feat_f if fl is not None else None)
        except Exception:  # noqa: BLE001  (a foreign cloud class with another constructor)
            filtered = PointClouds3D(pts_f, nrm_f, feat_f if fl is not None else None)
        keep_all = torch.cat(keeps)
        if filtered.isempty():
            return self._empty_fragments(N, dev, raster_settings), filtered
        kw = dict(kwargs)
        kw["verbose"] = True
        fragments, _, info = self._forward_masked(filtered, filtered, None, **kw)
        vis_ext = torch.zeros(keep_all.shape[0], dtype=torch.bool, device=dev)
        vis_ext[keep_all] = fragments.geometry[2].view(torch.bool) if fragments.geometry[2].dtype == torch.uint8 \
            else fragments.geometry[2].bool()
        self._store_visibility(point_clouds_filter, vis_ext, N, False, original_clouds)
        if kwargs.get("verbose", False):
            full = {}
            for k, v in info.items():   # rasterizer.py:655-662: per-point info scattered back over the un-filtered points
                full[k] = v.new_zeros((keep_all.shape[0],) + tuple(v.shape[1:]))
                full[k][keep_all] = v
            return fragments, filtered, full
        return fragments, filtered

    def forward(self, point_clouds, point_clouds_filter=None, **kwargs):
        raster_settings = kwargs.get("raster_settings", self.raster_settings)
        original_clouds = point_clouds
        if not point_clouds.isempty():
            point_clouds = self._apply_activation_filter(point_clouds, point_clouds_filter)
        if point_clouds.isempty():
            cameras = kwargs.get("cameras", self.cameras)
            return self._empty_fragments(cameras.R.shape[0], point_clouds.device, raster_settings), point_clouds
        if self.compacts(raster_settings):
            return self._forward_compacted(point_clouds, original_clouds, point_clouds_filter, **kwargs)
        return self._forward_masked(point_clouds, original_clouds, point_clouds_filter, **kwargs)

    def _forward_masked(self, point_clouds, original_clouds, point_clouds_filter, **kwargs):
        raster_settings = kwargs.get("raster_settings", self.raster_settings)
        a = self._prepare(point_clouds, **kwargs)
        N, shared, first_idx, num_points = a["N"], a["shared"], a["first_idx"], a["num_points"]

        pts_screen, ellipse, radii, scaler, cutoff, valid = _ProjectAndSetup.apply(
            a["world"], a["normals"], a["h"], a["M"], a["V"], a["znear"], a["zfar"], first_idx, num_points,
            raster_settings.image_size, raster_settings.cutoff_threshold, raster_settings.antialiasing_sigma,
            bool(raster_settings.backface_culling), shared, a["vr6"], a["frame_n"])

        idx, zbuf, qvalue_map, occ_map, visible = _EllipticalRasterizerWithVisibility.apply(
            pts_screen, ellipse, cutoff, radii, first_idx, num_points, raster_settings.depth_merging_threshold,
            raster_settings.image_size, raster_settings.points_per_pixel, raster_settings.bin_size,
            raster_settings.max_points_per_bin, raster_settings.radii_backward_scaler,
            raster_settings.clip_pts_grad)

        # the per-fragment scaler gather of rasterizer.py:631-633 is fused into the blend kernel: the fragments keep
        # the per-POINT scaler (`scaler_packed`) and materialise the (N,H,W,K) `scaler` only if it is read
        fragments = PointFragments(idx=idx, zbuf=zbuf, qvalue=qvalue_map, scaler=scaler, occupancy=occ_map,
                                   geometry=(pts_screen.detach(), radii, visible, first_idx, num_points))
        self._last_valid = valid
        self._store_visibility(point_clouds_filter, visible, N, shared, original_clouds)
        if kwargs.get("verbose", False):
            info = {"radii": radii, "ellipse_params": ellipse, "cutoff_threshold": cutoff, "scaler": scaler}
            return fragments, a["out_clouds"], info
        return fragments, a["out_clouds"]

    def _lean_plan(self, a, feats, st):
        """The ops.FusedPlan of this call's shape, or None when the inputs need the general (checking, converting) path."""
        from .ops import FusedPlan
        tensors = (a["world"], a["normals"], a["h"], a["M"], a["V"], a["znear"], a["zfar"], feats)
        if not all(FusedPlan.lean_input(t) for t in tensors) or st.points_per_pixel > 32 or \
                not FusedPlan.lean_input(a["first_idx"], torch.int64) or not FusedPlan.lean_input(a["num_points"], torch.int64) or \
                (a["vr6"] is not None and not (FusedPlan.lean_input(a["vr6"]) and FusedPlan.lean_input(a["frame_n"]))):
            return None
        dev, N, Pw = a["world"].device, a["N"], a["world"].shape[0]
        if a["normals"].shape != a["world"].shape or tuple(a["M"].shape) != (N, 4, 4) or tuple(a["V"].shape) != (N, 4, 4) or \
                a["znear"].numel() != N or a["zfar"].numel() != N or a["first_idx"].numel() != N:
            return None
        P = N * Pw if a["shared"] else Pw
        h = a["h"]
        per_point = 1 if (h.numel() == Pw and not (h.numel() == N and Pw == N)) else 0
        if not per_point and a["shared"] and N > 1 and h.numel() == P:
            per_point = 2          # one value per (camera, point) pair
        if feats.shape[0] != P or (not per_point and h.numel() != N) or P == 0:
            return None
        key = (dev, N, Pw, P, int(st.image_size), int(st.points_per_pixel), feats.shape[1], bool(a["shared"]), per_point,
               a["vr6"] is not None, bool(st.backface_culling), float(st.cutoff_threshold), float(st.antialiasing_sigma),
               float(st.depth_merging_threshold))
        plans = self.__dict__.setdefault("_plans", {})
        plan = plans.get(key)
        if plan is None:
            if len(plans) > 8:
                plans.clear()
            plan = plans[key] = FusedPlan(*key)
        return plan

    def _render_sharded(self, a, feats, st, part, point_clouds_filter, original_clouds, **kwargs):
        """`render_fused` on a row partition (`dss_amd.sharded.RowShardedRender`, multi-GPU): -> (image, fragments, clouds)
        with ``image`` the FULL (N,S,S,C+1) render (gathered from all ranks) or, with ``band_only=True``, this rank's rows
        (N,rows,S,C+1) for `dss_amd.distributed.band_image_loss`.  The fragments are those of the rank's rows."""
        from .sharded import RowShardedRender
        if st.points_per_pixel > 32:
            raise ValueError("a row-partitioned render needs points_per_pixel <= 32 (the fused kernels)")
        dev, N, Pw = a["world"].device, a["N"], a["world"].shape[0]
        P = N * Pw if a["shared"] else Pw
        S, K, C = int(st.image_size), int(st.points_per_pixel), int(feats.shape[1])
        if part.cyclic and C != 3:
            # the tile-row-cyclic variants of the backward are built for RGB features (include/dss_hip.h)
            if kwargs.get("row_partition_auto", False):
                from .distributed import RowPartition
                part = RowPartition(S, part.world_size, part.rank)          # contiguous equal bands instead
            else:
                raise ValueError("a tile-row-cyclic row partition needs 3 feature channels, got %d: use contiguous bands "
                                 "(RowPartition(S, world, rank) or row_partition='bands')" % C)
        from .sharded import choose_gradient_exchange
        band_only = bool(kwargs.get("band_only", False))
        gradient = kwargs.get("gradient_exchange", "auto")
        if gradient == "auto":   # (a replicated loss hands the owner form the full gradient: no alpha-plane exchange)
            gradient = choose_gradient_exchange(N, Pw, P, S, C, part.world_size, band_loss=band_only)
        group = kwargs.get("process_group", None)
        key = (dev, N, Pw, P, S, K, C, bool(a["shared"]), bool(st.backface_culling), float(st.cutoff_threshold),
               float(st.antialiasing_sigma), float(st.depth_merging_threshold), part.world_size, part.rank, part.cyclic,
               None if part.cyclic else tuple(part._bounds), gradient, id(group))
        engines = self.__dict__.setdefault("_sharded", {})
        engine = engines.get(key)
        if engine is None:
            if len(engines) > 4:
                engines.clear()
            engine = engines[key] = RowShardedRender(part, N, Pw, P, S, K, C, dev, a["shared"], st.cutoff_threshold,
                                                     st.antialiasing_sigma, st.depth_merging_threshold,
                                                     bool(st.backface_culling), group=group, gradient=gradient,
                                                     static_buffers=False)
        aux = (a["normals"], a["h"], a["M"], a["V"], a["znear"], a["zfar"], a["first_idx"], a["num_points"], a["vr6"],
               a["frame_n"], float(st.radii_backward_scaler), -1.0 if st.clip_pts_grad is None else float(st.clip_pts_grad),
               band_only, int(kwargs.get("order_refresh", getattr(self, "order_refresh", 0)) or 0))
        image = _RenderRowSharded.apply(a["world"], feats.contiguous(), engine, aux)
        f = engine.f
        visible = engine.vis_all.view(torch.bool)
        fragments = None
        if kwargs.get("want_fragments", True):
            fragments = PointFragments(idx=f["idx"], zbuf=f["zbuf"], qvalue=f["qvalue"], scaler=f["scaler"],
                                       occupancy=f["occupancy"],
                                       geometry=(f["pts_screen"], f["radii"], visible, a["first_idx"], a["num_points"]))
        self._store_visibility(point_clouds_filter, visible.clone(), a["N"], a["shared"], original_clouds)
        return image, fragments, a["out_clouds"]

    def render_fused(self, point_clouds, point_clouds_filter=None, **kwargs):
        """Rasterize AND blend in the fused kernels (``dss_render_forward`` / ``dss_render_backward``):
        -> ``(images (N,S,S,C+1), PointFragments, point_clouds)``.  Same values as ``forward`` + the
        renderer's blend; the autograd graph is one node, so gradients flow to the world points and the
        features only (a loss on ``fragments.zbuf`` needs the unfused path)."""
        original_clouds = point_clouds
        if kwargs.get("graphed", False) and point_clouds_filter is None and not kwargs.get("want_fragments", True):
            # graphed replay of the SAME call as last time (same tensor objects, cameras, settings, h: the steady state of
            # a training loop): nothing to prepare, nothing to check but the two addresses an optimiser could have replaced
            G = self.__dict__.get("_graphed")
            if G is not None and G.call_key is not None:
                pl, nl, fl = point_clouds.points_list(), point_clouds.normals_list(), point_clouds.features_list()
                k = G.call_key
                if len(pl) == 1 and pl[0] is k[0] and nl is not None and nl[0] is k[1] and fl is not None and fl[0] is k[2] \
                        and kwargs.get("Vrk_h", None) is k[3] and kwargs.get("cameras", self.cameras) is k[4] \
                        and kwargs.get("raster_settings", self.raster_settings) is k[5] and k[3]._version == k[6] \
                        and pl[0].data_ptr() == G.ptrs[0][0] and fl[0].data_ptr() == G.ptrs[9][0] \
                        and all(getattr(k[4], n, None) is t and getattr(t, "_version", 0) == v for n, t, v in k[7]) \
                        and k[8] == tuple(getattr(k[5], n, None) for n in PointsRasterizationSettings.__slots__):
                    return _RenderFusedGraphed.apply(pl[0], fl[0], G), None, point_clouds
        if not point_clouds.isempty():
            point_clouds = self._apply_activation_filter(point_clouds, point_clouds_filter)
        if point_clouds.isempty():  # like forward(): empty fragments, zero image
            st = kwargs.get("raster_settings", self.raster_settings)
            cameras = kwargs.get("cameras", self.cameras)
            frag = self._empty_fragments(cameras.R.shape[0], point_clouds.device, st)
            image = torch.zeros(tuple(frag.occupancy.shape) + (4,), device=point_clouds.device)
            return image, frag, point_clouds
        a = self._prepare(point_clouds, **kwargs)
        st = a["raster_settings"]
        feats = a["out_clouds"].features_packed()
        if feats.shape[1] > 8:
            raise ValueError("render_fused blends at most 8 feature channels, got %d (use the unfused forward() + "
                             "renderer for wider features)" % feats.shape[1])
        part = kwargs.get("row_partition", None)
        if part is not None:
            return self._render_sharded(a, feats, st, part, point_clouds_filter, original_clouds, **kwargs)
        lean = self._lean_plan(a, feats, st)
        # renderer-owned cached point order (large clouds; include/dss_hip.h DSS_WS_ORDER_*): refreshed every k-th call
        order_refresh = int(kwargs.get("order_refresh", getattr(self, "order_refresh", 0)) or 0)
        if lean is not None:
            lean.order_refresh = order_refresh
        if lean is not None and kwargs.get("graphed", False):
            inputs = (a["world"], a["normals"], a["h"], a["M"], a["V"], a["znear"], a["zfar"], a["first_idx"], a["num_points"],
                      feats, a["vr6"], a["frame_n"])
            radii_s = float(st.radii_backward_scaler)
            clip = -1.0 if st.clip_pts_grad is None else float(st.clip_pts_grad)
            G = self.__dict__.get("_graphed")
            if G is None or G.plan is not lean or G.ptrs != _GraphedRender.signature(inputs) or G.radii_s != radii_s \
                    or G.clip != clip:
                G = self._graphed = _GraphedRender(lean, inputs, radii_s, clip, a["shared"])
            # key of the steady-state shortcut at the top: only for one un-extended cloud whose tensors ARE the graph's inputs
            pl, nl, fl = point_clouds.points_list(), point_clouds.normals_list(), point_clouds.features_list()
            h_given, cams = kwargs.get("Vrk_h", None), kwargs.get("cameras", self.cameras)
            G.call_key = None
            if len(pl) == 1 and a["N"] == 1 and h_given is not None and nl is not None and fl is not None \
                    and pl[0] is a["world"] and fl[0] is feats and hasattr(cams, "R") and hasattr(cams, "T"):
                G.call_key = (pl[0], nl[0], fl[0], h_given, cams, st, h_given._version,
                              _camera_state(cams),
                              tuple(getattr(st, n, None) for n in PointsRasterizationSettings.__slots__))   # (settings mutate in place)
            image = _RenderFusedGraphed.apply(a["world"], feats, G)
            arena = G.arena
        elif lean is not None:
            aux = (a["normals"], a["h"], a["M"], a["V"], a["znear"], a["zfar"], a["first_idx"], a["num_points"], a["vr6"],
                   a["frame_n"], float(st.radii_backward_scaler), -1.0 if st.clip_pts_grad is None else float(st.clip_pts_grad),
                   a["shared"])
            image, arena = _RenderFusedLean.apply(a["world"], feats, lean, aux)
        if lean is not None:
            want = kwargs.get("want_fragments", True)
            fragments = None
            if want or (point_clouds_filter is not None and hasattr(point_clouds_filter, "set_filter")):
                visible = lean.view(arena, "visible").view(torch.bool)
                if want:
                    fragments = PointFragments(idx=lean.view(arena, "idx"), zbuf=lean.view(arena, "zbuf"),
                                               qvalue=lean.view(arena, "qvalue"), scaler=lean.view(arena, "scaler"),
                                               occupancy=lean.view(arena, "occupancy"),
                                               geometry=(lean.view(arena, "pts_screen"), lean.view(arena, "radii"), visible,
                                                         a["first_idx"], a["num_points"]))
                self._store_visibility(point_clouds_filter, visible, a["N"], a["shared"], original_clouds)
            return image, fragments, a["out_clouds"]
        outs = _RenderFused.apply(a["world"], feats.contiguous(), a["normals"], a["h"],
                                  a["M"], a["V"], a["znear"], a["zfar"], a["first_idx"], a["num_points"],
                                  st.image_size, st.points_per_pixel, st.cutoff_threshold, st.depth_merging_threshold,
                                  st.antialiasing_sigma, bool(st.backface_culling), a["shared"],
                                  st.radii_backward_scaler, st.clip_pts_grad, a["vr6"], a["frame_n"], order_refresh)
        image, idx, zbuf, qv, occ, scaler, pts_screen, radii, visible = outs
        fragments = PointFragments(idx=idx, zbuf=zbuf, qvalue=qv, scaler=scaler, occupancy=occ,
                                   geometry=(pts_screen, radii, visible, a["first_idx"], a["num_points"]))
        self._store_visibility(point_clouds_filter, visible, a["N"], a["shared"], original_clouds)
        return image, fragments, a["out_clouds"]


class _RenderRowSharded(autograd.Function):
    """The fused renderer with its image rows partitioned over the ranks of a process group (`dss_amd.sharded`), as ONE
    autograd node: forward = this rank's rows + the exchanges, backward = this rank's rows + the gradient exchange.  The
    gradients it returns are the sums over all ranks, identical on every rank."""

    @staticmethod
    def forward(ctx, world, features, engine, aux):
        normals, h, M, V, znear, zfar, first, num, vr6, frame_n, radii_s, clip, band_only, order_refresh = aux
        f = engine.forward(world, normals, h, M, V, znear, zfar, first, num, features, vr6, frame_n, order_refresh)
        # (private copies of what lives in the engine's exchange buffers: a second render before this one's backward --
        # Model.prune_points, an evaluation pass -- must not change what this node differentiates)
        vis = engine.start_exchange().clone()
        ctx.save_for_backward(world)
        ctx.engine, ctx.f, ctx.vis, ctx.aux = engine, f, vis, (M, V, first, num, radii_s, clip, not band_only)
        # band_only: the image all-gather stays in flight behind the loss and the backward (`engine.full_image()` waits for
        # it; the next forward does before it overwrites the send buffer)
        image = engine.band_image if band_only else engine.full_image()
        return image.clone(memory_format=torch.contiguous_format)

    @staticmethod
    def backward(ctx, g_image):
        (world,) = ctx.saved_tensors
        M, V, first, num, radii_s, clip, full = ctx.aux
        g_world, g_feat = ctx.engine.backward(g_image, radii_s, clip, world, M, V, first, num, f=ctx.f, vis_all=ctx.vis, full=full)
        return g_world.clone(), g_feat.clone(), None, None


class _GraphedRender:
    """``SurfaceSplattingRenderer(..., graphed=True)``: forward and backward of the fused renderer as two hipGraphs over
    static buffers -- per iteration the host issues two graph launches, one copy of the incoming image gradient and two
    small clones instead of ~10 kernel launches with their Python around them.  The graphs read the caller's tensors IN
    PLACE (parameters updated in place by an optimiser keep their address); a changed address or shape re-captures.
    Contract (as with any captured training step): ONE render in flight per renderer -- the image and the fragments of
    a forward are overwritten by the next forward; backward before rendering again."""

    def __init__(self, plan, inputs, radii_s, clip, shared):
        self.plan, self.radii_s, self.clip, self.shared = plan, radii_s, clip, shared
        self.call_key = None              # (see SurfaceSplatting.render_fused)
        self.inputs = inputs              # (world, normals, h, M, V, znear, zfar, first, num, feats, vr6, frame_n): kept alive
        self.ptrs = self.signature(inputs)
        world, normals, h, M, V, znear, zfar, first, num, feats, vr6, frame_n = inputs
        dev = world.device
        # The graphs bake the addresses of both workspaces into their launches: they are PRIVATE to this object (the cached
        # per-stream buffers of _lib are evicted / regrown by eager calls, and the capture stream's handle -- their cache key --
        # returns to PyTorch's stream pool), allocated outside the capture and alive as long as the graphs are.
        self.side = side = torch.cuda.Stream(device=dev)
        self.fwd_ws = torch.zeros(int(plan.fwd_ws_bytes), dtype=torch.uint8, device=dev)            # DSS_WS_CLEAN: zero-filled once
        self.bwd_ws = torch.empty(max(int(plan.bwd_ws_bytes), 256), dtype=torch.uint8, device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(2):   # warm-up on the capture stream
                arena = plan.forward(world, normals, h, M, V, znear, zfar, first, num, feats, vr6, frame_n, ws=self.fwd_ws)
            self.g_static = torch.zeros(plan.layout["image"][3], dtype=torch.float32, device=dev)
            fuse = self._fuse(world)
            plan.backward(arena, self.g_static, first, num, radii_s, clip, world if fuse else None, M if fuse else None,
                          ws=self.bwd_ws)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph_f = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph_f, stream=side):
            self.arena = plan.forward(world, normals, h, M, V, znear, zfar, first, num, feats, vr6, frame_n, ws=self.fwd_ws)
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph_b, stream=side):
            g_feat, g_pts = plan.backward(self.arena, self.g_static, first, num, radii_s, clip, world if fuse else None,
                                          M if fuse else None, ws=self.bwd_ws)
            if not fuse:
                g_pts = ops.project_backward(world, M, V, first, num, g_pts, plan.view(self.arena, "valid").view(torch.bool), shared)
            self.g_feat, self.g_world = g_feat, g_pts
        self.image = plan.image(self.arena)

    def _fuse(self, world):
        plan = self.plan
        widest = plan.N * plan.S * plan.S * max(plan.K, plan.C + 1) * 4
        return (not self.shared or plan.N == 1) and plan.C == 3 and world.shape[0] == plan.P and widest < (1 << 32) \
            and ops.backward_addr64() != 1

    @staticmethod
    def signature(inputs):
        return tuple((t.data_ptr(), t.shape[0]) if t is not None else None for t in inputs)


class _RenderFusedGraphed(autograd.Function):
    @staticmethod
    def forward(ctx, world, features, graphed):
        graphed.graph_f.replay()
        ctx.graphed = graphed
        return graphed.image.view(graphed.image.shape)

    @staticmethod
    def backward(ctx, g_image):
        G = ctx.graphed
        G.g_static.copy_(g_image)
        G.graph_b.replay()
        # the static gradient buffers themselves: AccumulateGrad copies a gradient it cannot take ownership of (these are
        # referenced by the graph object) or adds it into an existing .grad -- no clone here
        return G.g_world, G.g_feat, None


class _RenderFusedLean(autograd.Function):
    """`_RenderFused` with the host work of a call cut down (ops.FusedPlan): one arena instead of 13 output tensors, two
    autograd outputs instead of nine, prebuilt argument lists.  Same kernels, same numbers."""

    @staticmethod
    def forward(ctx, world, features, plan, aux):
        normals, h, M, V, znear, zfar, first, num, vr6, frame_n, radii_s, clip, shared = aux
        arena = plan.forward(world, normals, h, M, V, znear, zfar, first, num, features, vr6, frame_n)
        ctx.save_for_backward(world)
        ctx.plan, ctx.arena, ctx.aux = plan, arena, (M, V, first, num, radii_s, clip, shared)
        image = plan.image(arena)
        ctx.mark_non_differentiable(arena)
        return image, arena

    @staticmethod
    def backward(ctx, g_image, _g_arena=None):
        (world,) = ctx.saved_tensors
        plan, arena = ctx.plan, ctx.arena
        M, V, first, num, radii_s, clip, shared = ctx.aux
        if not g_image.is_contiguous():
            g_image = g_image.contiguous()
        widest = plan.N * plan.S * plan.S * max(plan.K, plan.C + 1) * 4
        fuse = (not shared or plan.N == 1) and plan.C == 3 and world.shape[0] == plan.P and widest < (1 << 32) \
            and ops.backward_addr64() != 1
        g_feat, g_pts = plan.backward(arena, g_image, first, num, radii_s, clip, world if fuse else None, M if fuse else None)
        if not fuse:
            g_pts = ops.project_backward(world, M, V, first, num, g_pts, plan.view(arena, "valid").view(torch.bool), shared)
        return g_pts, g_feat, None, None


class _RenderFused(autograd.Function):
    """One autograd node for the whole hot path: forward = dss_render_forward ([setup+bin] -> [fine+blend]),
    backward = dss_render_backward (+clip) -> dss_project_backward."""

    @staticmethod
    def forward(ctx, world, features, normals, h, M, V, znear, zfar, first_idx, num_points, image_size,
                points_per_pixel, cutoff, merge_thr, sigma, backface, shared, radii_s, clip, vr6=None,
                frame_normals=None, order_refresh=0):
        f = ops.render_forward(world, normals, h, M, V, znear, zfar, first_idx, num_points, features, image_size,
                               points_per_pixel, cutoff, merge_thr, sigma, backface, shared, vr6=vr6,
                               frame_normals=frame_normals, order_refresh=int(order_refresh))
        ctx.save_for_backward(world, M, V, first_idx, num_points, f["idx"], f["qvalue"], f["wsum"], f["scaler"],
                              f["pts_screen"], f["radii"], f["visible"], f["valid"])
        ctx.shared, ctx.radii_s = shared, float(radii_s)
        ctx.clip = -1.0 if clip is None else float(clip)
        outs = (f["image"], f["idx"], f["zbuf"], f["qvalue"], f["occupancy"], f["scaler"], f["pts_screen"],
                f["radii"], f["visible"])
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_image, *unused):
        world, M, V, first_idx, num_points, idx, qv, wsum, scaler, pts_screen, radii, visible, valid = ctx.saved_tensors
        # the projection rides in the gather's epilogue only on the 32-bit-offset variants of the kernel (every gathered
        # tensor below 4 GB and DSS_OPT_BACKWARD_ADDR64 off, raster_backward.hip): larger renders take the 64-bit gather
        # and the separate projection kernel instead of failing inside autograd
        widest = idx.numel() // idx.shape[-1] * max(idx.shape[-1], g_image.shape[-1]) * 4
        fuse = (not ctx.shared or M.shape[0] == 1) and g_image.shape[-1] == 4 and world.shape[0] == pts_screen.shape[0] \
            and widest < (1 << 32) and ops.backward_addr64() != 1
        g_feat, g_pts = ops.render_backward(g_image.contiguous(), idx, qv, wsum, scaler, pts_screen, radii, visible,
                                            first_idx, num_points, ctx.radii_s, ctx.clip,
                                            project=(world, M) if fuse else None)
        # clouds that are not shared between cameras: the projection backward ran in the gather's epilogue (g_pts is
        # already the world-space gradient); a shared cloud sums its cameras in the separate kernel
        g_world = g_pts if fuse else ops.project_backward(world, M, V, first_idx, num_points, g_pts, valid, ctx.shared)
        return (g_world, g_feat) + (None,) * 20
