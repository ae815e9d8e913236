"""Point shading: the MI355X drop-in for ``DSS.core.texture.LightingTexture`` and the light classes of
``DSS.core.lighting`` (texture.py:65-125, lighting.py:175-302).

``LightingTexture(cameras=..., lights=...).forward(pointclouds, shininess=64, **kwargs{lights, cameras, points_rgb})``
returns a point cloud whose features are ``rgb * (ambient + diffuse) + specular``; one fused HIP kernel each way
(``dss_phong_forward`` / ``dss_phong_backward``).  This is the path by which an RGB loss reaches the normals (and,
through point lights and the view direction, the positions).
"""
import torch
import torch.autograd as autograd

from . import ops
from .cloud import PointClouds3D, shared_cloud_ranges

__all__ = ["LightingTexture", "PointLights", "DirectionalLights"]


def _as_nl3(v, device):
    t = torch.as_tensor(v, dtype=torch.float32, device=device)
    if t.dim() == 1:
        t = t[None, None]
    elif t.dim() == 2:
        t = t[None]
    if t.dim() != 3 or t.shape[-1] != 3:
        raise ValueError("light properties must be (3,), (L,3) or (N,L,3), got %r" % (tuple(t.shape),))
    return t


class _Lights:
    """(N or 1, L, 3) ambient / diffuse / specular colours + location or direction (lighting.py:175-302)."""
    _vec = "location"

    def __init__(self, ambient_color=(((0.5, 0.5, 0.5),),), diffuse_color=(((0.3, 0.3, 0.3),),),
                 specular_color=(((0.2, 0.2, 0.2),),), device="cpu", **kwargs):
        self.device = torch.device(device)
        self.ambient_color = _as_nl3(ambient_color, self.device)
        self.diffuse_color = _as_nl3(diffuse_color, self.device)
        self.specular_color = _as_nl3(specular_color, self.device)
        setattr(self, self._vec, _as_nl3(kwargs.get(self._vec, ((0, 1, 0),)), self.device))
        for prop in ("diffuse_color", "specular_color", self._vec):
            if getattr(self, prop).dim() != 3:
                raise ValueError("%s must be an (N,L,3) tensor" % prop)

    def to(self, device):
        device = torch.device(device)
        if all(getattr(self, k).device == device or (device.index is None and getattr(self, k).device.type == device.type)
               for k in ("ambient_color", "diffuse_color", "specular_color", self._vec)):
            return self
        out = object.__new__(type(self))
        out.device = torch.device(device)
        for k in ("ambient_color", "diffuse_color", "specular_color", self._vec):
            setattr(out, k, getattr(self, k).to(device))
        return out

    def clone(self):
        out = object.__new__(type(self))
        out.device = self.device
        for k in ("ambient_color", "diffuse_color", "specular_color", self._vec):
            setattr(out, k, getattr(self, k).clone())
        return out

    def _packed(self, N):
        """-> ambient (N,3) summed over lights (texture.py:50-54), kd, ks, vec (N,L,3)."""
        L = max(self.diffuse_color.shape[1], self.specular_color.shape[1], getattr(self, self._vec).shape[1])
        ex = lambda t: t.expand(N, L, 3).contiguous()
        amb = self.ambient_color.sum(dim=1).expand(N, 3).contiguous()
        return amb, ex(self.diffuse_color), ex(self.specular_color), ex(getattr(self, self._vec))


class PointLights(_Lights):
    _vec = "location"

    def __init__(self, ambient_color=(((0.5, 0.5, 0.5),),), diffuse_color=(((0.3, 0.3, 0.3),),),
                 specular_color=(((0.2, 0.2, 0.2),),), location=(((0, 1, 0),),), device="cpu", **kwargs):
        super().__init__(ambient_color, diffuse_color, specular_color, device, location=location)


class DirectionalLights(_Lights):
    _vec = "direction"

    def __init__(self, ambient_color=(((0.5, 0.5, 0.5),),), diffuse_color=(((0.3, 0.3, 0.3),),),
                 specular_color=(((0.2, 0.2, 0.2),),), direction=(((0, 1, 0),),), device="cpu", **kwargs):
        super().__init__(ambient_color, diffuse_color, specular_color, device, direction=direction)


class _Phong(autograd.Function):
    @staticmethod
    def forward(ctx, world, normals, rgb, first, num, amb, kd, ks, vec, cam, point_lights, shininess, shared):
        out = ops.phong_forward(world, normals, rgb, first, num, amb, kd, ks, vec, point_lights, cam, shininess, shared)
        ctx.save_for_backward(world, normals, rgb, first, num, amb, kd, ks, vec, cam)
        ctx.cfg = (point_lights, shininess, shared)
        return out

    @staticmethod
    def backward(ctx, g):
        world, normals, rgb, first, num, amb, kd, ks, vec, cam = ctx.saved_tensors
        point_lights, shininess, shared = ctx.cfg
        gw, gn, gc = ops.phong_backward(g.contiguous(), world, normals, rgb, first, num, amb, kd, ks, vec, point_lights,
                                        cam, shininess, shared)
        return (gw, gn, gc) + (None,) * 10


class LightingTexture(torch.nn.Module):
    def __init__(self, device="cpu", cameras=None, lights=None, materials=None):
        super().__init__()
        self.lights = lights
        self.cameras = cameras

    def forward(self, pointclouds, shininess=64, **kwargs):
        if pointclouds.isempty():
            return pointclouds
        dev = pointclouds.device
        lights = kwargs.get("lights", self.lights).to(dev)
        cameras = kwargs.get("cameras", self.cameras).to(dev)
        N = len(cameras)
        shared = len(pointclouds) == 1 and N >= 1
        if not shared and len(pointclouds) != N:
            raise ValueError("need 1 or %d point clouds for %d cameras" % (N, N))
        world, normals = pointclouds.points_packed(), pointclouds.normals_packed()
        if normals is None:
            raise ValueError("LightingTexture needs point normals")
        out_clouds = pointclouds.extend(N) if shared and N > 1 else pointclouds
        points_rgb = kwargs.get("points_rgb", None)
        if points_rgb is None:
            feats = out_clouds.features_packed()
            points_rgb = feats[:, :3] if feats is not None else torch.ones((out_clouds.points_packed().shape[0], 3),
                                                                           device=dev)
        if points_rgb.shape[-1] != 3:
            raise ValueError("Expected points_rgb to be 3-channel, got %r" % (tuple(points_rgb.shape),))
        if shared:
            first, num = shared_cloud_ranges(N, world.shape[0], dev)
        else:
            first, num = pointclouds.cloud_to_packed_first_idx(), pointclouds.num_points_per_cloud()
        amb, kd, ks, vec = lights._packed(N)
        cam = cameras.get_camera_center().to(dev, torch.float32).reshape(-1, 3).expand(N, 3).contiguous()
        shaded = _Phong.apply(world, normals, points_rgb.contiguous(), first, num, amb, kd, ks, vec, cam,
                              isinstance(lights, PointLights), float(shininess), shared)
        lens = [p.shape[0] for p in out_clouds.points_list()]   # host-side sizes: no device synchronisation
        colored = PointClouds3D(out_clouds.points_list(), out_clouds.normals_list(), list(shaded.split(lens, 0)))
        return colored
