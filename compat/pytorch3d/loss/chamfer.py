"""Bidirectional chamfer distance between point sets (pytorch3d.loss.chamfer_distance interface)."""
import torch
import torch.nn.functional as F

from ..ops.knn import knn_gather, knn_points


def _handle(points, lengths, normals):
    if hasattr(points, "points_padded"):
        X = points.points_padded()
        lengths = points.num_points_per_cloud()
        normals = points.normals_padded()
    elif torch.is_tensor(points):
        if points.dim() != 3:
            raise ValueError("Expected points to be of shape (N, P, D)")
        X = points
        if lengths is not None and (lengths.dim() != 1 or lengths.shape[0] != X.shape[0]):
            raise ValueError("Expected lengths to be of shape (N,)")
        if lengths is None:
            lengths = torch.full((X.shape[0],), X.shape[1], dtype=torch.int64, device=points.device)
        if normals is not None and normals.dim() != 3:
            raise ValueError("Expected normals to be of shape (N, P, 3")
    else:
        raise ValueError("The input pointclouds should be either Pointclouds objects or torch.Tensor of shape (minibatch, "
                         "num_points, 3).")
    return X, lengths, normals


def chamfer_distance(x, y, x_lengths=None, y_lengths=None, x_normals=None, y_normals=None, weights=None,
                     batch_reduction="mean", point_reduction: str = "mean"):
    if batch_reduction is not None and batch_reduction not in ("mean", "sum"):
        raise ValueError('batch_reduction must be one of ["mean", "sum"] or None')
    if point_reduction not in ("mean", "sum"):
        raise ValueError('point_reduction must be one of ["mean", "sum"]')
    x, x_lengths, x_normals = _handle(x, x_lengths, x_normals)
    y, y_lengths, y_normals = _handle(y, y_lengths, y_normals)
    return_normals = x_normals is not None and y_normals is not None
    N, P1, D = x.shape
    P2 = y.shape[1]
    if y.shape[0] != N or y.shape[2] != D:
        raise ValueError("y does not have the correct shape.")
    is_x_heterogeneous = (x_lengths != P1).any()
    is_y_heterogeneous = (y_lengths != P2).any()
    x_mask = torch.arange(P1, device=x.device)[None] >= x_lengths[:, None]
    y_mask = torch.arange(P2, device=y.device)[None] >= y_lengths[:, None]
    if weights is not None:
        if weights.size(0) != N:
            raise ValueError("weights must be of shape (N,).")
        if not (weights >= 0).all():
            raise ValueError("weights cannot be negative.")
        if weights.sum() == 0.0:
            weights = weights.view(N, 1)
            if batch_reduction in ("mean", "sum"):
                return ((x.sum((1, 2)) * weights).sum() * 0.0, (x.sum((1, 2)) * weights).sum() * 0.0)
            return ((x.sum((1, 2)) * weights) * 0.0, (x.sum((1, 2)) * weights) * 0.0)
    x_nn = knn_points(x, y, lengths1=x_lengths, lengths2=y_lengths, K=1)
    y_nn = knn_points(y, x, lengths1=y_lengths, lengths2=x_lengths, K=1)
    cham_x = x_nn.dists[..., 0]
    cham_y = y_nn.dists[..., 0]
    if is_x_heterogeneous:
        cham_x = cham_x.masked_fill(x_mask, 0.0)
    if is_y_heterogeneous:
        cham_y = cham_y.masked_fill(y_mask, 0.0)
    if weights is not None:
        cham_x = cham_x * weights.view(N, 1)
        cham_y = cham_y * weights.view(N, 1)
    cham_norm_x = cham_norm_y = None
    if return_normals:
        x_normals_near = knn_gather(y_normals, x_nn.idx, y_lengths)[..., 0, :]
        y_normals_near = knn_gather(x_normals, y_nn.idx, x_lengths)[..., 0, :]
        cham_norm_x = 1 - torch.abs(F.cosine_similarity(x_normals, x_normals_near, dim=2, eps=1e-6))
        cham_norm_y = 1 - torch.abs(F.cosine_similarity(y_normals, y_normals_near, dim=2, eps=1e-6))
        if is_x_heterogeneous:
            cham_norm_x = cham_norm_x.masked_fill(x_mask, 0.0)
        if is_y_heterogeneous:
            cham_norm_y = cham_norm_y.masked_fill(y_mask, 0.0)
        if weights is not None:
            cham_norm_x = cham_norm_x * weights.view(N, 1)
            cham_norm_y = cham_norm_y * weights.view(N, 1)
    cham_x = cham_x.sum(1)
    cham_y = cham_y.sum(1)
    if return_normals:
        cham_norm_x = cham_norm_x.sum(1)
        cham_norm_y = cham_norm_y.sum(1)
    if point_reduction == "mean":
        cham_x = cham_x / x_lengths
        cham_y = cham_y / y_lengths
        if return_normals:
            cham_norm_x = cham_norm_x / x_lengths
            cham_norm_y = cham_norm_y / y_lengths
    if batch_reduction is not None:
        cham_x = cham_x.sum()
        cham_y = cham_y.sum()
        if return_normals:
            cham_norm_x = cham_norm_x.sum()
            cham_norm_y = cham_norm_y.sum()
        if batch_reduction == "mean":
            div = weights.sum() if weights is not None else N
            cham_x = cham_x / div
            cham_y = cham_y / div
            if return_normals:
                cham_norm_x = cham_norm_x / div
                cham_norm_y = cham_norm_y / div
    cham_dist = cham_x + cham_y
    cham_normals = cham_norm_x + cham_norm_y if return_normals else None
    return cham_dist, cham_normals
