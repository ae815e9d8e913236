"""Directional / point lights with pytorch3d.renderer.lighting's interface (Phong terms per point)."""
import torch
import torch.nn.functional as F

from .utils import TensorProperties, convert_to_tensors_and_broadcast


def diffuse(normals, color, direction) -> torch.Tensor:
    """color * max(0, n . l); normals (..., 3), color / direction (N, 3) or broadcastable to normals"""
    if not torch.is_tensor(color):
        normals, color, direction = convert_to_tensors_and_broadcast(normals, color, direction, device=normals.device)
    if normals.shape != direction.shape:
        direction = direction.view((-1,) + (1,) * (normals.dim() - 2) + (3,)) if direction.dim() == 2 and \
            direction.shape[0] in (1, normals.shape[0]) and normals.dim() > 2 else direction
        color = color.view((-1,) + (1,) * (normals.dim() - 2) + (3,)) if color.dim() == 2 and normals.dim() > 2 else color
    normals = F.normalize(normals, p=2, dim=-1, eps=1e-6)
    direction = F.normalize(direction, p=2, dim=-1, eps=1e-6)
    angle = F.relu(torch.sum(normals * direction, dim=-1))
    return color * angle[..., None]


def specular(points, normals, direction, color, camera_position, shininess) -> torch.Tensor:
    """color * max(0, r . v)^shininess with r the reflection of the light direction about the normal"""
    if points.shape != normals.shape:
        raise ValueError("Expected points and normals to have the same shape: got %r, %r" % (points.shape, normals.shape))
    if not torch.is_tensor(shininess):
        shininess = torch.tensor(shininess, dtype=points.dtype, device=points.device)
    extra = (1,) * (points.dim() - 2)
    if direction.dim() == 2 and points.dim() > 2:
        direction = direction.view((-1,) + extra + (3,))
    if color.dim() == 2 and points.dim() > 2:
        color = color.view((-1,) + extra + (3,))
    if camera_position.dim() == 2 and points.dim() > 2:
        camera_position = camera_position.view((-1,) + extra + (3,))
    if shininess.dim() == 1 and points.dim() > 2:
        shininess = shininess.view((-1,) + extra)
    normals = F.normalize(normals, p=2, dim=-1, eps=1e-6)
    direction = F.normalize(direction, p=2, dim=-1, eps=1e-6)
    cos_angle = torch.sum(normals * direction, dim=-1)
    mask = (cos_angle > 0).to(torch.float32)
    view_direction = F.normalize(camera_position - points, p=2, dim=-1, eps=1e-6)
    reflect_direction = -direction + 2 * (cos_angle[..., None] * normals)
    alpha = F.relu(torch.sum(view_direction * reflect_direction, dim=-1)) * mask
    return color * torch.pow(alpha, shininess)[..., None]


class DirectionalLights(TensorProperties):
    def __init__(self, ambient_color=((0.5, 0.5, 0.5),), diffuse_color=((0.3, 0.3, 0.3),),
                 specular_color=((0.2, 0.2, 0.2),), direction=((0, 1, 0),), device="cpu"):
        super().__init__(device=device, ambient_color=ambient_color, diffuse_color=diffuse_color,
                         specular_color=specular_color, direction=direction)
        _validate_light_properties(self)
        if self.direction.shape[-1] != 3:
            raise ValueError("Expected direction to have shape (N, 3); got %r" % repr(self.direction.shape))

    def clone(self):
        other = self.__class__(device=self.device)
        return super().clone(other)

    def diffuse(self, normals, points=None) -> torch.Tensor:
        return diffuse(normals=normals, color=self.diffuse_color, direction=self.direction)

    def specular(self, normals, points, camera_position, shininess) -> torch.Tensor:
        return specular(points=points, normals=normals, color=self.specular_color, direction=self.direction,
                        camera_position=camera_position, shininess=shininess)


class PointLights(TensorProperties):
    def __init__(self, ambient_color=((0.5, 0.5, 0.5),), diffuse_color=((0.3, 0.3, 0.3),),
                 specular_color=((0.2, 0.2, 0.2),), location=((0, 1, 0),), device="cpu"):
        super().__init__(device=device, ambient_color=ambient_color, diffuse_color=diffuse_color,
                         specular_color=specular_color, location=location)
        _validate_light_properties(self)
        if self.location.shape[-1] != 3:
            raise ValueError("Expected location to have shape (N, 3); got %r" % repr(self.location.shape))

    def clone(self):
        other = self.__class__(device=self.device)
        return super().clone(other)

    def _direction(self, points):
        loc = self.location
        if loc.dim() == 2 and points.dim() > 2:
            loc = loc.view((-1,) + (1,) * (points.dim() - 2) + (3,))
        return loc - points

    def diffuse(self, normals, points) -> torch.Tensor:
        return diffuse(normals=normals, color=self.diffuse_color, direction=self._direction(points))

    def specular(self, normals, points, camera_position, shininess) -> torch.Tensor:
        return specular(points=points, normals=normals, color=self.specular_color, direction=self._direction(points),
                        camera_position=camera_position, shininess=shininess)


def _validate_light_properties(obj):
    for n in ("ambient_color", "diffuse_color", "specular_color"):
        t = getattr(obj, n)
        if t.shape[-1] != 3:
            raise ValueError("Expected %s to have shape (N, 3); got %r" % (n, t.shape))
