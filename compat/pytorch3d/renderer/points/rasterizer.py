"""PointsRasterizationSettings / PointFragments / PointsRasterizer with pytorch3d's interface.  The base rasterizer only
carries `cameras` + `raster_settings` and the world -> NDC `transform` (the reference subclasses it:
DSS/core/rasterizer.py:102) -- its forward (pytorch3d's disc rasterizer) is not provided."""
from typing import NamedTuple, Optional

import torch
import torch.nn as nn


class PointFragments(NamedTuple):
    idx: torch.Tensor
    zbuf: torch.Tensor
    dists: torch.Tensor


class PointsRasterizationSettings:
    __slots__ = ["image_size", "radius", "points_per_pixel", "bin_size", "max_points_per_bin"]

    def __init__(self, image_size: int = 256, radius: float = 0.01, points_per_pixel: int = 8,
                 bin_size: Optional[int] = None, max_points_per_bin: Optional[int] = None):
        self.image_size = image_size
        self.radius = radius
        self.points_per_pixel = points_per_pixel
        self.bin_size = bin_size
        self.max_points_per_bin = max_points_per_bin


class PointsRasterizer(nn.Module):
    def __init__(self, cameras=None, raster_settings=None):
        super().__init__()
        if raster_settings is None:
            raster_settings = PointsRasterizationSettings()
        self.cameras = cameras
        self.raster_settings = raster_settings

    def transform(self, point_clouds, **kwargs):
        """world -> NDC (x, y) with VIEW-space depth as z"""
        cameras = kwargs.get("cameras", self.cameras)
        if cameras is None:
            raise ValueError("Cameras must be specified either at initialization or in the forward pass of PointsRasterizer")
        pts_world = point_clouds.points_padded()
        eps = kwargs.get("eps", None)
        pts_view = cameras.get_world_to_view_transform(**kwargs).transform_points(pts_world, eps=eps)
        pts_screen = cameras.get_projection_transform(**kwargs).transform_points(pts_view, eps=eps)
        pts_screen[..., 2] = pts_view[..., 2]
        return point_clouds.update_padded(pts_screen)

    def to(self, device):
        if self.cameras is not None:
            self.cameras = self.cameras.to(device)
        return self

    def forward(self, point_clouds, **kwargs):
        raise NotImplementedError("pytorch3d's disc rasterizer is not part of this compatibility layer")
