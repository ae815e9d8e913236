"""Normal / local-frame estimation with pytorch3d.ops.points_normals' interface (PCA of the K-neighbourhoods)."""
import torch

from .utils import convert_pointclouds_to_tensor, get_point_covariances


def estimate_pointcloud_local_coord_frames(pointclouds, neighborhood_size: int = 50, disambiguate_directions: bool = True):
    """-> curvatures (N, P, 3) ascending eigenvalues, local_coord_frames (N, P, 3, 3) with the eigenvectors as COLUMNS"""
    points_padded, num_points = convert_pointclouds_to_tensor(pointclouds)
    ba, N, dim = points_padded.shape
    if dim != 3:
        raise ValueError("The pointclouds argument has to be of shape (minibatch, N, 3)")
    if (num_points <= neighborhood_size).any():
        raise ValueError("The neighborhood_size argument has to be strictly smaller than the size of each point cloud.")
    cov, knns = get_point_covariances(points_padded, num_points, neighborhood_size)
    curvatures, local_coord_frames = torch.linalg.eigh(cov)
    if disambiguate_directions:
        n = _disambiguate_vector_directions(points_padded, knns, local_coord_frames[:, :, :, 0])
        z = _disambiguate_vector_directions(points_padded, knns, local_coord_frames[:, :, :, 2])
        y = torch.cross(z, n, dim=2)
        local_coord_frames = torch.stack((n, y, z), dim=3)
    return curvatures, local_coord_frames


def estimate_pointcloud_normals(pointclouds, neighborhood_size: int = 50, disambiguate_directions: bool = True):
    _, local_coord_frames = estimate_pointcloud_local_coord_frames(pointclouds, neighborhood_size=neighborhood_size,
                                                                   disambiguate_directions=disambiguate_directions)
    return local_coord_frames[:, :, :, 0]


def _disambiguate_vector_directions(pcl, knns, vecs):
    """flip each vector so that the majority of the neighbours lies on its positive side (Tombari et al., SHOT)"""
    df = knns - pcl[:, :, None]
    proj = (vecs[:, :, None] * df).sum(3)
    n_pos = (proj > 0).to(proj.dtype).sum(2, keepdim=True)
    flip = (n_pos < (0.5 * knns.shape[2])).to(proj.dtype)
    return (1.0 - 2.0 * flip) * vecs
