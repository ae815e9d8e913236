"""pytorch3d.ops.utils helpers the reference imports."""
import torch


def is_pointclouds(pcl) -> bool:
    return hasattr(pcl, "points_padded") and hasattr(pcl, "num_points_per_cloud")


def convert_pointclouds_to_tensor(pcl):
    """Pointclouds or (N, P, D) tensor -> (padded tensor, num_points (N,))"""
    if is_pointclouds(pcl):
        X = pcl.points_padded()
        num_points = pcl.num_points_per_cloud()
    elif torch.is_tensor(pcl):
        X = pcl
        num_points = X.shape[1] * torch.ones(X.shape[0], device=X.device, dtype=torch.int64)
    else:
        raise ValueError("The inputs X, Y should be either Pointclouds objects or tensors.")
    return X, num_points


def eyes(dim: int, N: int, device=None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    return torch.eye(dim, device=device, dtype=dtype)[None].repeat(N, 1, 1)


def wmean(x, weight=None, dim=-2, keepdim: bool = True, eps: float = 1e-9):
    args = {"dim": dim, "keepdim": keepdim}
    if weight is None:
        return x.mean(**args)
    return (x * weight[..., None]).sum(**args) / weight[..., None].sum(**args).clamp(eps)


def get_point_covariances(points_padded, num_points_per_cloud, neighborhood_size: int):
    """per-point covariance of the K-neighbourhood about its mean: (N, P, 3, 3), and the neighbours (N, P, K, 3)"""
    from .knn import knn_points
    k_nearest_neighbors = knn_points(points_padded, points_padded, lengths1=num_points_per_cloud,
                                     lengths2=num_points_per_cloud, K=neighborhood_size, return_nn=True).knn
    pt_mean = k_nearest_neighbors.mean(2, keepdim=True)
    central_diff = k_nearest_neighbors - pt_mean
    per_pt_cov = central_diff.unsqueeze(4) * central_diff.unsqueeze(3)
    per_pt_cov = per_pt_cov.mean(2)
    return per_pt_cov, k_nearest_neighbors
