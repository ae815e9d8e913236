from .compositor import AlphaCompositor, NormWeightedCompositor
from .rasterize_points import kMaxPointsPerBin, rasterize_points
from .rasterizer import PointFragments, PointsRasterizationSettings, PointsRasterizer
from .renderer import PointsRenderer
