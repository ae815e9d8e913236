"""Minimal camera objects with pytorch3d's conventions, sufficient for the splatting hot path.

pytorch3d is a third-party dependency of the reference that is absent here; only the few methods
DSS/core/rasterizer.py calls on a camera object are restated (from pytorch3d 0.2.5-0.4.0's
published behaviour, [third party, not under /root/reference]):

* row-vector convention: ``X_view = X_world @ R + T``; 4x4 matrices act as ``p_h @ M``;
* the camera looks down +Z with +X to the left and +Y up; NDC x,y in [-1, 1];
* ``FoVPerspectiveCameras``: ``x_ndc = s*x/z / aspect``, ``y_ndc = s*y/z``, ``w = z`` with
  ``s = 1/tan(fov/2)``; ``z_ndc = zfar/(zfar-znear) - zfar*znear/((zfar-znear) z)``;
* ``get_world_to_view_transform()`` / ``get_full_projection_transform()`` return objects with
  ``get_matrix()`` and ``transform_points()`` (used at rasterizer.py:138-145, 188-189, 465-466).

A genuine pytorch3d camera can be passed to ``dss_amd`` instead: only these methods are used.
"""
import math
from typing import Optional

import torch


class Transform3d:
    """(N,4,4) row-vector transform: ``p_h @ M``."""

    def __init__(self, matrix: torch.Tensor):
        self._matrix = matrix if matrix.dim() == 3 else matrix[None]

    def get_matrix(self) -> torch.Tensor:
        return self._matrix

    def compose(self, other: "Transform3d") -> "Transform3d":
        return Transform3d(self._matrix @ other._matrix)

    def transform_points(self, points: torch.Tensor, eps: Optional[float] = None) -> torch.Tensor:
        pts = points if points.dim() == 3 else points[None]
        ones = torch.ones_like(pts[..., :1])
        ph = torch.cat([pts, ones], dim=-1) @ self._matrix
        denom = ph[..., 3:]
        if eps is not None:
            sign = denom.sign() + (denom == 0).type_as(denom)
            denom = sign * denom.abs().clamp(min=eps)
        out = ph[..., :3] / denom
        return out if points.dim() == 3 else out[0]

    def transform_normals(self, normals: torch.Tensor) -> torch.Tensor:
        nrm = normals if normals.dim() == 3 else normals[None]
        mat = self._matrix[:, :3, :3]
        out = nrm @ torch.inverse(mat).transpose(1, 2)
        return out if normals.dim() == 3 else out[0]


def _as_batch(v, n, device, dtype=torch.float32):
    t = torch.as_tensor(v, dtype=dtype, device=device).reshape(-1)
    return t.expand(n) if t.numel() == 1 else t


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees: bool = True,
                           at=((0.0, 0.0, 0.0),), up=((0.0, 1.0, 0.0),), device="cpu"):
    """pytorch3d.renderer.look_at_view_transform -> (R (N,3,3), T (N,3))."""
    dist, elev, azim = (torch.as_tensor(x, dtype=torch.float32, device=device).reshape(-1) for x in (dist, elev, azim))
    n = max(dist.numel(), elev.numel(), azim.numel())
    dist, elev, azim = (x.expand(n) if x.numel() == 1 else x for x in (dist, elev, azim))
    if degrees:
        elev, azim = elev * (math.pi / 180.0), azim * (math.pi / 180.0)
    at = torch.as_tensor(at, dtype=torch.float32, device=device).reshape(-1, 3).expand(n, 3)
    up = torch.as_tensor(up, dtype=torch.float32, device=device).reshape(-1, 3).expand(n, 3)
    cam = torch.stack([dist * torch.cos(elev) * torch.sin(azim), dist * torch.sin(elev),
                       dist * torch.cos(elev) * torch.cos(azim)], dim=-1) + at
    z_axis = torch.nn.functional.normalize(at - cam, eps=1e-5)
    x_axis = torch.nn.functional.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = torch.nn.functional.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    close = torch.isclose(x_axis, torch.zeros_like(x_axis), atol=5e-3).all(dim=1, keepdim=True)
    if close.any():
        x_axis = torch.where(close, torch.nn.functional.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5), x_axis)
    R = torch.cat([x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]], dim=1).transpose(1, 2)
    T = -torch.bmm(R.transpose(1, 2), cam[:, :, None])[:, :, 0]
    return R, T


class FoVPerspectiveCameras:
    """Subset of pytorch3d.renderer.FoVPerspectiveCameras used by the hot path."""

    def __init__(self, znear=1.0, zfar=100.0, aspect_ratio=1.0, fov=60.0, degrees: bool = True,
                 R: Optional[torch.Tensor] = None, T: Optional[torch.Tensor] = None, device="cpu"):
        self.device = torch.device(device)
        self.R = (torch.eye(3)[None] if R is None else torch.as_tensor(R, dtype=torch.float32)).to(self.device)
        self.T = (torch.zeros(1, 3) if T is None else torch.as_tensor(T, dtype=torch.float32)).to(self.device)
        if self.R.dim() == 2:
            self.R = self.R[None]
        if self.T.dim() == 1:
            self.T = self.T[None]
        n = max(self.R.shape[0], self.T.shape[0])
        self.R = self.R.expand(n, 3, 3)
        self.T = self.T.expand(n, 3)
        self.znear = _as_batch(znear, n, self.device)
        self.zfar = _as_batch(zfar, n, self.device)
        self.aspect_ratio = _as_batch(aspect_ratio, n, self.device)
        self.fov = _as_batch(fov, n, self.device)
        self.degrees = degrees

    def __len__(self):
        return self.R.shape[0]

    def to(self, device):
        device = torch.device(device)
        if all(getattr(self, k).device == device or (device.index is None and getattr(self, k).device.type == device.type)
               for k in self._STATE):
            return self  # already there: keep the object (and its matrix cache)
        other = FoVPerspectiveCameras.__new__(FoVPerspectiveCameras)
        other.__dict__.update(self.__dict__)
        other.__dict__.pop("_matrix_cache", None)
        other.device = device
        for k in ("R", "T", "znear", "zfar", "aspect_ratio", "fov"):
            setattr(other, k, getattr(self, k).to(device))
        return other

    _STATE = ("R", "T", "znear", "zfar", "aspect_ratio", "fov")

    def _cached(self, name, build):
        """Matrices are rebuilt only when a camera tensor was replaced or modified in place (tensor identity +
        version counter; the cache keeps the tensors alive so an address cannot be recycled).  A training loop asks
        for the same matrices every iteration, and each rebuild is ~20 tiny GPU launches (0.3 ms of host time)."""
        state = tuple(getattr(self, k) for k in self._STATE)
        key = tuple(t._version for t in state) + (self.degrees,)
        cache = self.__dict__.setdefault("_matrix_cache", {})
        hit = cache.get(name)
        if hit is not None and hit[1] == key and all(a is b for a, b in zip(hit[0], state)):
            return hit[2]
        value = build()
        cache[name] = (state, key, value)
        return value

    def get_world_to_view_transform(self, **kwargs) -> Transform3d:
        if "R" not in kwargs and "T" not in kwargs:
            return self._cached("view", self._world_to_view)
        return self._world_to_view(**kwargs)

    def _world_to_view(self, **kwargs) -> Transform3d:
        R = kwargs.get("R", self.R)
        T = kwargs.get("T", self.T)
        n = R.shape[0]
        m = torch.zeros(n, 4, 4, dtype=torch.float32, device=R.device)
        m[:, :3, :3] = R
        m[:, 3, :3] = T
        m[:, 3, 3] = 1.0
        return Transform3d(m)

    def get_projection_transform(self, **kwargs) -> Transform3d:
        return self._cached("projection", self._projection)

    def _projection(self) -> Transform3d:
        n = len(self)
        dev = self.R.device
        fov = self.fov * (math.pi / 180.0) if self.degrees else self.fov
        tan_half = torch.tan(fov / 2)
        K = torch.zeros(n, 4, 4, dtype=torch.float32, device=dev)
        max_y = tan_half * self.znear
        min_y = -max_y
        max_x = max_y * self.aspect_ratio
        min_x = -max_x
        K[:, 0, 0] = 2.0 * self.znear / (max_x - min_x)
        K[:, 1, 1] = 2.0 * self.znear / (max_y - min_y)
        K[:, 0, 2] = (max_x + min_x) / (max_x - min_x)
        K[:, 1, 2] = (max_y + min_y) / (max_y - min_y)
        K[:, 3, 2] = 1.0
        K[:, 2, 2] = self.zfar / (self.zfar - self.znear)
        K[:, 2, 3] = -(self.zfar * self.znear) / (self.zfar - self.znear)
        return Transform3d(K.transpose(1, 2).contiguous())

    def get_full_projection_transform(self, **kwargs) -> Transform3d:
        if "R" not in kwargs and "T" not in kwargs:
            return self._cached("full", lambda: self._world_to_view().compose(self._projection()))
        return self._world_to_view(**kwargs).compose(self.get_projection_transform())

    def transform_points(self, points, eps: Optional[float] = None, **kwargs):
        return self.get_full_projection_transform(**kwargs).transform_points(points, eps=eps)

    def get_camera_center(self, **kwargs):
        R = kwargs.get("R", self.R)
        T = kwargs.get("T", self.T)
        return -torch.bmm(R, T[:, :, None])[:, :, 0]


class CameraSampler:
    """Random look-at cameras in batches, mirroring `DSS.core.camera.CameraSampler` (DSS/core/camera.py:6-73): distances
    uniform in ``distance_range`` (sorted descending when ``sort_distance``), azimuth U[-180, 180), elevation U[-90, 90),
    look-at point U[-0.05, 0.05)^3 -- drawn from torch's global generator in the reference's order, so the same seed gives
    the same cameras -- then `look_at_view_transform`; iterating yields ``camera_type(R=..., T=..., **camera_params)`` for
    ``num_cams_batch`` views at a time."""

    def __init__(self, num_cams_total, num_cams_batch, distance_range=((5.0, 10.0),), sort_distance=True, return_cams=True,
                 camera_type=None, camera_params=None):
        self.num_cams_batch = num_cams_batch
        self.num_cams_total = num_cams_total
        self.sort_distance = sort_distance
        self.camera_type = FoVPerspectiveCameras if camera_type is None else camera_type
        self.camera_params = {} if camera_params is None else camera_params
        distance_range = torch.as_tensor(distance_range, dtype=torch.float32).reshape(-1, 2)
        distance_scale = distance_range[:, -1] - distance_range[:, 0]
        distances = torch.rand(num_cams_total) * distance_scale + distance_range[:, 0]
        if sort_distance:
            distances, _ = distances.sort(descending=True)
        azim = torch.rand(num_cams_total) * 360 - 180
        elev = torch.rand(num_cams_total) * 180 - 90
        at = torch.rand((num_cams_total, 3)) * 0.1 - 0.05
        self.distances, self.azim, self.elev, self.at = distances, azim, elev, at
        self.R, self.T = look_at_view_transform(distances, elev, azim, at=at, degrees=True)
        self._idx = 0

    def __len__(self):
        return (self.R.shape[0] + self.num_cams_batch - 1) // self.num_cams_batch

    def __iter__(self):
        return self

    def __next__(self):
        if self._idx >= len(self):
            raise StopIteration
        start = self._idx * self.num_cams_batch
        end = min(start + self.num_cams_batch, self.R.shape[0])
        cameras = self.camera_type(R=self.R[start:end], T=self.T[start:end], **self.camera_params)
        self._idx += 1
        return cameras
