from .knn import knn_gather, knn_points
from .packed_to_padded import packed_to_padded, padded_to_packed
from .points_normals import estimate_pointcloud_local_coord_frames, estimate_pointcloud_normals
from .sample_points_from_meshes import sample_points_from_meshes
from .utils import convert_pointclouds_to_tensor, eyes, get_point_covariances, is_pointclouds, wmean
