from . import cameras, compositing, lighting, utils
from .cameras import (CamerasBase, FoVOrthographicCameras, FoVPerspectiveCameras, OpenGLOrthographicCameras,
                      OpenGLPerspectiveCameras, OrthographicCameras, PerspectiveCameras, SfMOrthographicCameras,
                      SfMPerspectiveCameras, camera_position_from_spherical_angles, get_world_to_view_transform,
                      look_at_rotation, look_at_view_transform)
from .compositing import alpha_composite, norm_weighted_sum, weighted_sum
from .lighting import DirectionalLights, PointLights, diffuse, specular
from .points import (AlphaCompositor, NormWeightedCompositor, PointFragments, PointsRasterizationSettings, PointsRasterizer,
                     PointsRenderer, rasterize_points)
from .utils import TensorProperties, convert_to_tensors_and_broadcast, format_tensor
