"""Point-cloud regularisers of the training iteration, mirroring DSS/training/losses.py.

`ProjectionLoss` and `RepulsionLoss` (reference losses.py:281-392, :395-492, both on `SurfaceLoss` :145-278) keep the
reference's constructor arguments, call signature (``loss(point_clouds, points_filter=..., rebuild_knn=True)``),
reduction handling (`BaseLoss` :24-62) and results; the Trainer builds them with ``reduction='mean',
filter_scale=2.0, knn_k=12`` (trainer.py:134-137).  The arithmetic runs in three HIP kernels on the packed neighbour
lists of `dss_knn_points` (dss_amd/csrc/regularizers.hip) instead of ~40 padded torch tensors; the neighbour search
itself replaces `pytorch3d.ops.knn_points`.  As in the reference, only the points receive a gradient (every weight is
computed under no_grad there).
"""
from typing import Optional

import torch
from torch import autograd

from . import neighbours, ops


class _Neighbourhood:
    """Self-query neighbour lists of a batch of clouds (the reference's `knn_tree`, losses.py:155-179), packed."""

    def __init__(self, point_clouds, K: int):
        self.first = point_clouds.cloud_to_packed_first_idx()
        self.num = point_clouds.num_points_per_cloud()
        self.K = int(K)
        self.dists, self.idx = neighbours.self_knn(point_clouds.points_packed(), self.first, self.num,
                                                   [p.shape[0] for p in point_clouds.points_list()], self.K)

    def matches(self, point_clouds) -> bool:
        num = point_clouds.num_points_per_cloud()
        return num.shape == self.num.shape and bool(torch.equal(num, self.num))


def _packed_mask(mask, point_clouds) -> Optional[torch.Tensor]:
    """(N, Pmax) padded or (P,) packed bool mask -> packed (P,) bool.  A mask with more rows than clouds (one row per
    camera of a shared cloud) is OR-ed over the rows like losses.py:203-206."""
    if mask is None:
        return None
    sizes = [p.shape[0] for p in point_clouds.points_list()]   # host-side sizes: no device synchronisation
    P = sum(sizes)
    if mask.dim() == 1:
        if mask.numel() != P:
            raise ValueError("Incompatible point clouds ({} points) and mask {}".format(P, tuple(mask.shape)))
        return mask.bool()
    if mask.shape[0] != len(point_clouds):
        if len(point_clouds) == 1 and mask.shape[0] > 1:
            mask = mask.any(dim=0, keepdim=True)
        else:
            raise ValueError("Incompatible point clouds {} and mask {}".format(len(point_clouds), tuple(mask.shape)))
    if len(sizes) == 1:
        return mask[0, : sizes[0]].bool()
    return torch.cat([mask[b, :n] for b, n in enumerate(sizes)]).bool()


class _Projection(autograd.Function):
    @staticmethod
    def forward(ctx, points, mollified, dists, idx, visible, first, num, sigma):
        loss, _ = ops.projection_loss(points, mollified, dists, idx, visible, first, num, sigma)
        ctx.save_for_backward(points, mollified, dists, idx, first, num)
        ctx.visible, ctx.sigma = visible, sigma
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        points, mollified, dists, idx, first, num = ctx.saved_tensors
        _, grad = ops.projection_loss(points, mollified, dists, idx, ctx.visible, first, num, ctx.sigma,
                                      grad_loss=grad_loss.contiguous(), want_loss=False, want_grad=True)
        return (grad,) + (None,) * 7


class _Repulsion(autograd.Function):
    @staticmethod
    def forward(ctx, points, mollified, idx, first, num, sigma, filter_scale):
        loss, _ = ops.repulsion_loss(points, mollified, idx, first, num, sigma, filter_scale)
        ctx.save_for_backward(points, mollified, idx, first, num)
        ctx.sigma, ctx.filter_scale = sigma, filter_scale
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        points, mollified, idx, first, num = ctx.saved_tensors
        _, grad = ops.repulsion_loss(points, mollified, idx, first, num, ctx.sigma, ctx.filter_scale,
                                     grad_loss=grad_loss.contiguous(), want_loss=False, want_grad=True)
        return (grad,) + (None,) * 6


class SurfaceLoss(torch.nn.Module):
    """Shared state of the two regularisers (reference `SurfaceLoss` + `BaseLoss`)."""

    def __init__(self, reduction: str = "mean", knn_k: int = 33, filter_scale: float = 1.0, sharpness_sigma: float = 0.75):
        super().__init__()
        self.reduction = reduction
        self.channel_dim = None
        self.knn_tree = None
        self.knn_k = knn_k
        neighbours.request_lists(knn_k)   # the renderer's kNN for `h` will now produce lists this loss can reuse
        self.filter_scale = filter_scale
        self.sharpness_sigma = sharpness_sigma

    def _reduce(self, loss, reduction=None):  # losses.py:42-52
        reduction = reduction or self.reduction
        if reduction == "none":
            return loss
        if reduction == "sum":
            return torch.sum(loss)
        if reduction == "mean":
            return torch.mean(loss)
        raise ValueError("Invalid reduction method ({})".format(reduction))

    def forward(self, *args, **kwargs):  # losses.py:54-61
        reduction = kwargs.pop("reduction", self.reduction)
        self.channel_dim = kwargs.pop("channel_dim", self.channel_dim)
        loss = self.compute(*args, **kwargs)
        if self.channel_dim is not None:
            loss = torch.sum(loss, dim=self.channel_dim)
        return self._reduce(loss, reduction=reduction)

    def _neighbourhood(self, point_clouds, rebuild_knn, kwargs):
        self.sharpness_sigma = kwargs.get("sharpness_sigma", self.sharpness_sigma)
        self.filter_scale = kwargs.get("filter_scale", self.filter_scale)
        self.knn_tree = kwargs.get("knn_tree", self.knn_tree)
        if self.knn_tree is not None and not isinstance(self.knn_tree, _Neighbourhood):
            raise TypeError("knn_tree must be the packed neighbourhood of a dss_amd loss (its `.knn_tree`), not a padded "
                            "pytorch3d result; pass rebuild_knn=True to search again")
        if rebuild_knn or self.knn_tree is None or not self.knn_tree.matches(point_clouds):
            self.knn_tree = _Neighbourhood(point_clouds, self.knn_k)
        return self.knn_tree

    @staticmethod
    def _mollified(point_clouds, nb, points_filter):
        keep = None
        if points_filter is not None:
            vis, inm = getattr(points_filter, "visibility", None), getattr(points_filter, "inmask", None)
            if vis is not None and inm is not None:
                # losses.py:198-206: AND first (same view), then OR over the views of a shared cloud
                both = (vis.bool() & inm.bool()) if vis.shape == inm.shape else None
                keep = _packed_mask(both, point_clouds) if both is not None else \
                    _packed_mask(vis, point_clouds) & _packed_mask(inm, point_clouds)
        return ops.mollify_normals(point_clouds.normals_packed().detach(), nb.dists, nb.idx, keep, nb.first, nb.num)


class ProjectionLoss(SurfaceLoss):
    """Weighted squared distance of every point to the planes of its neighbours (reference :281-392), (P,) before the
    reduction."""

    def compute(self, point_clouds, points_filter=None, rebuild_knn=False, **kwargs):
        nb = self._neighbourhood(point_clouds, rebuild_knn, kwargs)
        mollified = self._mollified(point_clouds, nb, points_filter)
        visible = None if points_filter is None else _packed_mask(points_filter.visibility, point_clouds)
        return _Projection.apply(point_clouds.points_packed(), mollified, nb.dists, nb.idx, visible, nb.first, nb.num,
                                 float(self.sharpness_sigma))


class RepulsionLoss(SurfaceLoss):
    """exp(-|tangential offset to the weighted neighbourhood|) per coordinate (reference :395-492), (P,3) before the
    reduction."""

    def compute(self, point_clouds, points_filter=None, rebuild_knn=True, **kwargs):
        nb = self._neighbourhood(point_clouds, rebuild_knn, kwargs)
        mollified = self._mollified(point_clouds, nb, points_filter)
        return _Repulsion.apply(point_clouds.points_packed(), mollified, nb.idx, nb.first, nb.num,
                                float(self.sharpness_sigma), float(self.filter_scale))


class _ImageLoss(autograd.Function):
    """total, loss_dr_rgb, loss_dr_silhouette, IoU term; only the total is differentiable (the others are what the
    Trainer logs)."""

    @staticmethod
    def forward(ctx, rgba, img, mask_img, lambda_rgb, lambda_silhouette):
        losses, sums = ops.image_loss_forward(rgba, img, mask_img, lambda_rgb, lambda_silhouette)
        ctx.save_for_backward(rgba, img, mask_img, sums)
        ctx.lambdas = (lambda_rgb, lambda_silhouette)
        total, rest = losses[0], losses[1:]
        ctx.mark_non_differentiable(rest)
        return total, rest

    @staticmethod
    def backward(ctx, grad_total, _grad_rest):
        rgba, img, mask_img, sums = ctx.saved_tensors
        grad = ops.image_loss_backward(rgba, img, mask_img, ctx.lambdas[0], ctx.lambdas[1], sums,
                                       grad_total=grad_total.contiguous())
        return grad, None, None, None, None


def calc_dr_loss(rgba_pred, img, mask_img, lambda_dr_rgb: float = 1.0, lambda_dr_silhouette: float = 1.0):
    """The image loss of `Trainer.calc_dr_loss` (trainer.py:332-372) on the renderer's (N,H,W,4) output
    (``img_pred = rgba[..., :3]``, ``mask_img_pred = rgba[..., 3]``): masked L1 on RGB + silhouette L1 + 0.01 IoU, as
    one reduction pass and, in backward, one gradient pass (dss_amd/csrc/image_loss.hip).

    ``img`` (N,H,W,3) float (a permuted NCHW view is fine, as in trainer.py:306), ``mask_img`` (N,H,W) or (N,1,H,W).
    Returns the Trainer's dictionary entries: ``loss`` (differentiable), ``loss_dr_rgb``, ``loss_dr_silhouette``, plus
    ``loss_iou`` -- device scalars, no host synchronisation."""
    if mask_img.dtype != torch.float32:
        mask_img = mask_img.float()
    total, rest = _ImageLoss.apply(rgba_pred, img, mask_img, float(lambda_dr_rgb), float(lambda_dr_silhouette))
    return {"loss": total, "loss_dr_rgb": rest[0], "loss_dr_silhouette": rest[1], "loss_iou": rest[2]}
