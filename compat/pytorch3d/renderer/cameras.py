"""Cameras with pytorch3d.renderer.cameras' interface and conventions (row-vector matrices; NDC with +x left, +y up,
+z into the screen; FoV cameras map z in [znear, zfar] to [0, 1])."""
import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from ..transforms import Rotate, Transform3d, Translate
from .utils import TensorProperties, convert_to_tensors_and_broadcast

_R = torch.eye(3)[None]
_T = torch.zeros(1, 3)


class CamerasBase(TensorProperties):
    def get_projection_transform(self, **kwargs):
        raise NotImplementedError()

    def unproject_points(self, xy_depth, world_coordinates: bool = True, **kwargs):
        raise NotImplementedError()

    def get_camera_center(self, **kwargs) -> torch.Tensor:
        """the world point that maps to the view origin: x_view = x_world R + T  =>  C = -T R^-1.  pytorch3d inverts the composed
        4 x 4 world-to-view transform and reads its last row; the reference's texture asks for it AFTER
        `cameras.gather_props(packed_to_cloud_idx)`, i.e. for one camera per POINT (8 x 99,790 of them at configs[2]): a batch
        of 798k 4 x 4 LU inverses and 4 x 4 products -- 16 ms of rocSOLVER / hipBLASLt kernels per iteration of train_mvr.py,
        more than half of its GPU time.  Same value in closed form, elementwise: the columns of R^-1 are the cross products
        of R's rows over the determinant."""
        R = kwargs.get("R", self.R)
        T = kwargs.get("T", self.T)
        self.R = R
        self.T = T
        if R.dim() == 2:
            R = R[None]
        if T.dim() == 1:
            T = T[None]
        r0, r1, r2 = R[:, 0, :], R[:, 1, :], R[:, 2, :]
        c0, c1, c2 = torch.cross(r1, r2, dim=1), torch.cross(r2, r0, dim=1), torch.cross(r0, r1, dim=1)
        det = (r0 * c0).sum(1, keepdim=True)
        if T.shape[0] != R.shape[0]:
            T = T.expand(R.shape[0], -1) if T.shape[0] == 1 else T
            c0, c1, c2, det = (v.expand(T.shape[0], -1) if v.shape[0] == 1 else v for v in (c0, c1, c2, det))
        return -torch.stack([(T * c0).sum(1), (T * c1).sum(1), (T * c2).sum(1)], dim=1) / det

    def get_world_to_view_transform(self, **kwargs) -> Transform3d:
        R = kwargs.get("R", self.R)
        T = kwargs.get("T", self.T)
        self.R = R
        self.T = T
        return get_world_to_view_transform(R=R, T=T)

    def get_full_projection_transform(self, **kwargs) -> Transform3d:
        self.R = kwargs.get("R", self.R)
        self.T = kwargs.get("T", self.T)
        world_to_view_transform = self.get_world_to_view_transform(R=self.R, T=self.T)
        view_to_proj_transform = self.get_projection_transform(**kwargs)
        return world_to_view_transform.compose(view_to_proj_transform)

    def transform_points(self, points, eps: Optional[float] = None, **kwargs) -> torch.Tensor:
        world_to_proj_transform = self.get_full_projection_transform(**kwargs)
        return world_to_proj_transform.transform_points(points, eps=eps)

    def transform_points_screen(self, points, image_size, eps: Optional[float] = None, **kwargs) -> torch.Tensor:
        ndc = self.transform_points(points, eps=eps, **kwargs)
        if not torch.is_tensor(image_size):
            image_size = torch.tensor(image_size, dtype=torch.int64, device=points.device)
        image_size = image_size.view(-1, 2)  # (height, width) per pytorch3d >= 0.4? (width, height) in 0.2; square here
        w = image_size[:, 1].to(ndc.dtype).view(-1, 1)
        h = image_size[:, 0].to(ndc.dtype).view(-1, 1)
        x = (w - 1.0) / 2.0 * (1.0 - ndc[..., 0])
        y = (h - 1.0) / 2.0 * (1.0 - ndc[..., 1])
        return torch.stack([x, y, ndc[..., 2]], dim=-1)

    def clone(self):
        cam_type = type(self)
        other = cam_type(device=self.device)
        return super().clone(other)

    def _unproject(self, xy_depth, world_coordinates, scaled_depth_input, **kwargs):
        if world_coordinates:
            to_ndc = self.get_full_projection_transform(**kwargs.copy())
        else:
            to_ndc = self.get_projection_transform(**kwargs.copy())
        return to_ndc.inverse().transform_points(xy_depth)


def get_world_to_view_transform(R=_R, T=_T) -> Transform3d:
    if T.shape[0] != R.shape[0]:
        raise ValueError("Expected R, T to have the same batch dimension; got %r, %r" % (R.shape[0], T.shape[0]))
    if T.dim() != 2 or T.shape[1:] != (3,):
        raise ValueError("Expected T to have shape (N, 3); got %r" % repr(T.shape))
    if R.dim() != 3 or R.shape[1:] != (3, 3):
        raise ValueError("Expected R to have shape (N, 3, 3); got %r" % repr(R.shape))
    T_ = Translate(T, device=T.device)
    R_ = Rotate(R, device=R.device)
    return R_.compose(T_)


class FoVPerspectiveCameras(CamerasBase):
    def __init__(self, znear=1.0, zfar=100.0, aspect_ratio=1.0, fov=60.0, degrees: bool = True, R=_R, T=_T, K=None,
                 device="cpu"):
        super().__init__(device=device, znear=znear, zfar=zfar, aspect_ratio=aspect_ratio, fov=fov, R=R, T=T, K=K)
        self.degrees = degrees

    def compute_projection_matrix(self, znear, zfar, fov, aspect_ratio, degrees: bool) -> torch.Tensor:
        K = torch.zeros((self._N, 4, 4), device=self.device, dtype=torch.float32)
        ones = torch.ones((self._N), dtype=torch.float32, device=self.device)
        if degrees:
            fov = (math.pi / 180) * fov
        if not torch.is_tensor(fov):
            fov = torch.tensor(fov, device=self.device)
        tanHalfFov = torch.tan((fov / 2))
        max_y = tanHalfFov * znear
        min_y = -max_y
        max_x = max_y * aspect_ratio
        min_x = -max_x
        z_sign = 1.0
        K[:, 0, 0] = 2.0 * znear / (max_x - min_x)
        K[:, 1, 1] = 2.0 * znear / (max_y - min_y)
        K[:, 0, 2] = (max_x + min_x) / (max_x - min_x)
        K[:, 1, 2] = (max_y + min_y) / (max_y - min_y)
        K[:, 3, 2] = z_sign * ones
        K[:, 2, 2] = z_sign * zfar / (zfar - znear)
        K[:, 2, 3] = -(zfar * znear) / (zfar - znear)
        return K

    def get_projection_transform(self, **kwargs) -> Transform3d:
        K = kwargs.get("K", self.K)
        if K is not None:
            if K.shape != (self._N, 4, 4):
                raise ValueError("Expected K to have shape of (%r, 4, 4)" % self._N)
        else:
            K = self.compute_projection_matrix(kwargs.get("znear", self.znear), kwargs.get("zfar", self.zfar),
                                               kwargs.get("fov", self.fov), kwargs.get("aspect_ratio", self.aspect_ratio),
                                               kwargs.get("degrees", self.degrees))
        transform = Transform3d(device=self.device)
        transform._matrix = K.transpose(1, 2).contiguous()
        return transform

    def unproject_points(self, xy_depth, world_coordinates: bool = True, scaled_depth_input: bool = False, **kwargs):
        if world_coordinates:
            to_ndc_transform = self.get_full_projection_transform(**kwargs.copy())
        else:
            to_ndc_transform = self.get_projection_transform(**kwargs.copy())
        if scaled_depth_input:
            xy_sdepth = xy_depth
        else:
            K_matrix = self.get_projection_transform(**kwargs.copy()).get_matrix()
            unsqueeze_shape = [1] * xy_depth.dim()
            unsqueeze_shape[0] = K_matrix.shape[0]
            f1 = K_matrix[:, 2, 2].reshape(unsqueeze_shape)
            f2 = K_matrix[:, 3, 2].reshape(unsqueeze_shape)
            sdepth = (f1 * xy_depth[..., 2:3] + f2) / xy_depth[..., 2:3]
            xy_sdepth = torch.cat((xy_depth[..., 0:2], sdepth), dim=-1)
        return to_ndc_transform.inverse().transform_points(xy_sdepth)


def OpenGLPerspectiveCameras(*args, **kwargs):  # the pre-0.3 name
    return FoVPerspectiveCameras(*args, **kwargs)


class FoVOrthographicCameras(CamerasBase):
    def __init__(self, znear=1.0, zfar=100.0, max_y=1.0, min_y=-1.0, max_x=1.0, min_x=-1.0,
                 scale_xyz=((1.0, 1.0, 1.0),), R=_R, T=_T, K=None, device="cpu"):
        super().__init__(device=device, znear=znear, zfar=zfar, max_y=max_y, min_y=min_y, max_x=max_x, min_x=min_x,
                         scale_xyz=scale_xyz, R=R, T=T, K=K)

    def compute_projection_matrix(self, znear, zfar, max_x, min_x, max_y, min_y, scale_xyz) -> torch.Tensor:
        K = torch.zeros((self._N, 4, 4), dtype=torch.float32, device=self.device)
        ones = torch.ones((self._N), dtype=torch.float32, device=self.device)
        z_sign = +1.0
        K[:, 0, 0] = (2.0 / (max_x - min_x)) * scale_xyz[:, 0]
        K[:, 1, 1] = (2.0 / (max_y - min_y)) * scale_xyz[:, 1]
        K[:, 0, 3] = -(max_x + min_x) / (max_x - min_x)
        K[:, 1, 3] = -(max_y + min_y) / (max_y - min_y)
        K[:, 3, 3] = ones
        K[:, 2, 2] = z_sign * (1.0 / (zfar - znear)) * scale_xyz[:, 2]
        K[:, 2, 3] = -znear / (zfar - znear)
        return K

    def get_projection_transform(self, **kwargs) -> Transform3d:
        K = kwargs.get("K", self.K)
        if K is not None:
            if K.shape != (self._N, 4, 4):
                raise ValueError("Expected K to have shape of (%r, 4, 4)" % self._N)
        else:
            K = self.compute_projection_matrix(kwargs.get("znear", self.znear), kwargs.get("zfar", self.zfar),
                                               kwargs.get("max_x", self.max_x), kwargs.get("min_x", self.min_x),
                                               kwargs.get("max_y", self.max_y), kwargs.get("min_y", self.min_y),
                                               kwargs.get("scale_xyz", self.scale_xyz))
        transform = Transform3d(device=self.device)
        transform._matrix = K.transpose(1, 2).contiguous()
        return transform

    def unproject_points(self, xy_depth, world_coordinates: bool = True, scaled_depth_input: bool = False, **kwargs):
        if world_coordinates:
            to_ndc_transform = self.get_full_projection_transform(**kwargs.copy())
        else:
            to_ndc_transform = self.get_projection_transform(**kwargs.copy())
        if scaled_depth_input:
            xy_sdepth = xy_depth
        else:
            K = self.get_projection_transform(**kwargs).get_matrix()
            unsqueeze_shape = [1] * K.dim()
            unsqueeze_shape[0] = K.shape[0]
            mid_z = K[:, 3, 2].reshape(unsqueeze_shape)
            scale = K[:, 2, 2].reshape(unsqueeze_shape)
            scaled_depth = scale * xy_depth[..., 2:3] + mid_z
            xy_sdepth = torch.cat((xy_depth[..., :2], scaled_depth), dim=-1)
        return to_ndc_transform.inverse().transform_points(xy_sdepth)


def OpenGLOrthographicCameras(*args, **kwargs):
    return FoVOrthographicCameras(*args, **kwargs)


def _focal_pp(focal_length, principal_point):
    if focal_length.dim() == 1 or focal_length.shape[-1] == 1:
        fx = fy = focal_length.reshape(-1)
    else:
        fx, fy = focal_length.unbind(1)
    px, py = principal_point.unbind(1)
    return fx, fy, px, py


class PerspectiveCameras(CamerasBase):
    """NDC-space pinhole: x' = fx X / Z + px, y' = fy Y / Z + py, z' = 1 / Z"""

    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), R=_R, T=_T, K=None, device="cpu",
                 image_size=((-1, -1),)):
        super().__init__(device=device, focal_length=focal_length, principal_point=principal_point, R=R, T=T, K=K,
                         image_size=image_size)

    def get_projection_transform(self, **kwargs) -> Transform3d:
        K = kwargs.get("K", self.K)
        if K is None:
            fx, fy, px, py = _focal_pp(kwargs.get("focal_length", self.focal_length),
                                       kwargs.get("principal_point", self.principal_point))
            n = max(fx.shape[0], px.shape[0])
            K = fx.new_zeros(n, 4, 4)
            K[:, 0, 0] = fx
            K[:, 1, 1] = fy
            K[:, 0, 2] = px
            K[:, 1, 2] = py
            K[:, 3, 2] = 1.0
            K[:, 2, 3] = 1.0
        transform = Transform3d(device=self.device)
        transform._matrix = K.transpose(1, 2).contiguous()
        return transform

    def unproject_points(self, xy_depth, world_coordinates: bool = True, **kwargs):
        if world_coordinates:
            to_ndc_transform = self.get_full_projection_transform(**kwargs)
        else:
            to_ndc_transform = self.get_projection_transform(**kwargs)
        xy_inv_depth = torch.cat((xy_depth[..., :2], 1.0 / xy_depth[..., 2:3]), dim=-1)
        return to_ndc_transform.inverse().transform_points(xy_inv_depth)


def SfMPerspectiveCameras(*args, **kwargs):
    return PerspectiveCameras(*args, **kwargs)


class OrthographicCameras(CamerasBase):
    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), R=_R, T=_T, K=None, device="cpu",
                 image_size=((-1, -1),)):
        super().__init__(device=device, focal_length=focal_length, principal_point=principal_point, R=R, T=T, K=K,
                         image_size=image_size)

    def get_projection_transform(self, **kwargs) -> Transform3d:
        K = kwargs.get("K", self.K)
        if K is None:
            fx, fy, px, py = _focal_pp(kwargs.get("focal_length", self.focal_length),
                                       kwargs.get("principal_point", self.principal_point))
            n = max(fx.shape[0], px.shape[0])
            K = fx.new_zeros(n, 4, 4)
            K[:, 0, 0] = fx
            K[:, 1, 1] = fy
            K[:, 0, 3] = px
            K[:, 1, 3] = py
            K[:, 2, 2] = 1.0
            K[:, 3, 3] = 1.0
        transform = Transform3d(device=self.device)
        transform._matrix = K.transpose(1, 2).contiguous()
        return transform

    def unproject_points(self, xy_depth, world_coordinates: bool = True, **kwargs):
        if world_coordinates:
            to_ndc_transform = self.get_full_projection_transform(**kwargs)
        else:
            to_ndc_transform = self.get_projection_transform(**kwargs)
        return to_ndc_transform.inverse().transform_points(xy_depth)


def SfMOrthographicCameras(*args, **kwargs):
    return OrthographicCameras(*args, **kwargs)


def camera_position_from_spherical_angles(distance, elevation, azimuth, degrees: bool = True, device="cpu") -> torch.Tensor:
    dist, elev, azim = convert_to_tensors_and_broadcast(distance, elevation, azimuth, device=device)
    if degrees:
        elev = math.pi / 180.0 * elev
        azim = math.pi / 180.0 * azim
    x = dist * torch.cos(elev) * torch.sin(azim)
    y = dist * torch.sin(elev)
    z = dist * torch.cos(elev) * torch.cos(azim)
    camera_position = torch.stack([x, y, z], dim=1)
    if camera_position.dim() == 0:
        camera_position = camera_position.view(1, -1)
    return camera_position.view(-1, 3)


def look_at_rotation(camera_position, at=((0, 0, 0),), up=((0, 1, 0),), device="cpu") -> torch.Tensor:
    """(N, 3, 3) world-to-view rotation in the row-vector convention: columns are the camera's x, y, z axes"""
    camera_position, at, up = convert_to_tensors_and_broadcast(camera_position, at, up, device=device)
    for t, n in zip([camera_position, at, up], ["camera_position", "at", "up"]):
        if t.shape[-1] != 3:
            raise ValueError("Expected arg %s to have shape (N, 3); got %r" % (n, t.shape))
    z_axis = F.normalize(at - camera_position, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        replacement = F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5)
        x_axis = torch.where(is_close, replacement, x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1)
    return R.transpose(1, 2)


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees: bool = True, eye: Optional[Sequence] = None,
                           at=((0, 0, 0),), up=((0, 1, 0),), device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    if eye is not None:
        eye, at, up = convert_to_tensors_and_broadcast(eye, at, up, device=device)
        C = eye
    else:
        dist, elev, azim, at, up = convert_to_tensors_and_broadcast(dist, elev, azim, at, up, device=device)
        C = camera_position_from_spherical_angles(dist, elev, azim, degrees=degrees, device=device) + at
    R = look_at_rotation(C, at, up, device=device)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T
