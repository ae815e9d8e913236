"""PointsRenderer (pytorch3d.renderer.points.renderer interface): rasterizer -> weights 1 - d^2 / r^2 -> compositor."""
import torch
import torch.nn as nn


class PointsRenderer(nn.Module):
    def __init__(self, rasterizer, compositor):
        super().__init__()
        self.rasterizer = rasterizer
        self.compositor = compositor

    def to(self, device):
        self.rasterizer = self.rasterizer.to(device)
        self.compositor = self.compositor.to(device) if self.compositor is not None else None
        return self

    def forward(self, point_clouds, **kwargs) -> torch.Tensor:
        fragments = self.rasterizer(point_clouds, **kwargs)
        r = self.rasterizer.raster_settings.radius
        dists2 = fragments.dists.permute(0, 3, 1, 2)
        weights = 1 - dists2 / (r * r)
        images = self.compositor(fragments.idx.long().permute(0, 3, 1, 2), weights,
                                 point_clouds.features_packed().permute(1, 0), **kwargs)
        return images.permute(0, 2, 3, 1)
