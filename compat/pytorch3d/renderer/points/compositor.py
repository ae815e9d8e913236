"""AlphaCompositor / NormWeightedCompositor modules (pytorch3d.renderer.points.compositor interface)."""
import torch
import torch.nn as nn

from ..compositing import alpha_composite, norm_weighted_sum


class _Base(nn.Module):
    def __init__(self, background_color=None):
        super().__init__()
        self.background_color = background_color

    def _background(self, images, fragments, kwargs):
        bg = kwargs.get("background_color", self.background_color)
        if bg is None:
            return images
        if not torch.is_tensor(bg):
            bg = torch.tensor(bg, dtype=images.dtype, device=images.device)
        empty = (fragments[:, 0] < 0)[:, None]            # (N, 1, H, W)
        C = images.shape[1]
        if bg.numel() == C - 1:                           # alpha channel of the background = 1
            bg = torch.cat([bg, bg.new_ones(1)])
        return torch.where(empty, bg.view(1, -1, 1, 1).expand_as(images), images)


class AlphaCompositor(_Base):
    def forward(self, fragments, alphas, ptclds, **kwargs) -> torch.Tensor:
        return self._background(alpha_composite(fragments, alphas, ptclds), fragments, kwargs)


class NormWeightedCompositor(_Base):
    def forward(self, fragments, alphas, ptclds, **kwargs) -> torch.Tensor:
        return self._background(norm_weighted_sum(fragments, alphas, ptclds), fragments, kwargs)
