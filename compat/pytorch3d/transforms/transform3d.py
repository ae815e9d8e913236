"""Lazy 4x4 transforms in pytorch3d's ROW-VECTOR convention (points are rows: p' = p @ M; the translation is the last
row).  Interface of pytorch3d.transforms.Transform3d / Translate / Scale / Rotate / RotateAxisAngle."""
import math
from typing import Optional

import torch


def _bmm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """batched matmul with broadcasting of a batch dimension of 1 (and of a missing one)"""
    if a.dim() == 2:
        a = a[None]
    if b.dim() == 2:
        b = b[None]
    if len(a) != len(b):
        if len(a) != 1 and len(b) != 1:
            raise ValueError("Expected batch dim for bmm to be equal or 1; got %r, %r" % (a.shape, b.shape))
        if len(a) == 1:
            a = a.expand(len(b), -1, -1)
        if len(b) == 1:
            b = b.expand(len(a), -1, -1)
    return a.bmm(b)


class Transform3d:
    def __init__(self, dtype: torch.dtype = torch.float32, device="cpu", matrix: Optional[torch.Tensor] = None):
        if matrix is None:
            self._matrix = torch.eye(4, dtype=dtype, device=device).view(1, 4, 4)
        else:
            if matrix.dim() not in (2, 3):
                raise ValueError('"matrix" has to be a 2- or a 3-dimensional tensor.')
            if matrix.shape[-2] != 4 or matrix.shape[-1] != 4:
                raise ValueError('"matrix" has to be a tensor of shape (minibatch, 4, 4)')
            dtype = matrix.dtype
            device = matrix.device
            self._matrix = matrix.view(-1, 4, 4)
        self._transforms = []
        self._lu = None
        self.device = torch.device(device) if not isinstance(device, torch.device) else device
        self.dtype = dtype

    def __len__(self) -> int:
        return self.get_matrix().shape[0]

    def __getitem__(self, index):
        if isinstance(index, int):
            index = [index]
        return self.__class__.__new__(Transform3d)._init_from(self.get_matrix()[index])

    def _init_from(self, matrix):
        Transform3d.__init__(self, matrix=matrix)
        return self

    def compose(self, *others):
        out = Transform3d(dtype=self.dtype, device=self.device)
        out._matrix = self._matrix.clone()
        for other in others:
            if not isinstance(other, Transform3d):
                raise ValueError("Only possible to compose Transform3d objects; got %s" % type(other))
        out._transforms = self._transforms + list(others)
        return out

    def get_matrix(self) -> torch.Tensor:
        composed = self._matrix.clone()
        for other in self._transforms:
            composed = _bmm(composed, other.get_matrix())
        return composed

    def _get_matrix_inverse(self) -> torch.Tensor:
        # same LU inverse as torch.inverse, without its host-side check of the singularity flag: that check waits for the GPU
        # (2 ms per iteration of train_mvr.py at configs[2], where get_camera_center() comes here once per render)
        return torch.linalg.inv_ex(self._matrix).inverse

    def inverse(self, invert_composed: bool = False):
        tinv = Transform3d(dtype=self.dtype, device=self.device)
        if invert_composed:
            tinv._matrix = torch.linalg.inv_ex(self.get_matrix()).inverse
        else:
            i_matrix = self._get_matrix_inverse()
            if len(self._transforms) > 0:
                tinv._transforms = [t.inverse() for t in reversed(self._transforms)]
                last = Transform3d(dtype=self.dtype, device=self.device)
                last._matrix = i_matrix
                tinv._transforms.append(last)
            else:
                tinv._matrix = i_matrix
        return tinv

    def stack(self, *others):
        transforms = [self] + list(others)
        out = Transform3d(dtype=self.dtype, device=self.device)
        out._matrix = torch.cat([t.get_matrix() for t in transforms], dim=0)
        return out

    def transform_points(self, points: torch.Tensor, eps: Optional[float] = None) -> torch.Tensor:
        points_batch = points.clone()
        if points_batch.dim() == 2:
            points_batch = points_batch[None]
        if points_batch.dim() != 3:
            raise ValueError("Expected points to have dim = 2 or dim = 3: got shape %r" % (tuple(points.shape),))
        N, P, _ = points_batch.shape
        ones = torch.ones(N, P, 1, dtype=points.dtype, device=points.device)
        points_batch = torch.cat([points_batch, ones], dim=2)
        composed = self.get_matrix()
        points_out = _bmm(points_batch, composed)
        denom = points_out[..., 3:]
        if eps is not None:
            denom_sign = denom.sign() + (denom == 0.0).type_as(denom)
            denom = denom_sign * torch.clamp(denom.abs(), eps)
        points_out = points_out[..., :3] / denom
        if points_out.shape[0] == 1 and points.dim() == 2:
            points_out = points_out.reshape(points.shape)
        return points_out

    def transform_normals(self, normals: torch.Tensor) -> torch.Tensor:
        if normals.dim() not in (2, 3):
            raise ValueError("Expected normals to have dim = 2 or dim = 3: got shape %r" % (tuple(normals.shape),))
        composed = self.get_matrix()
        mat = composed[:, :3, :3]
        normals_out = _bmm(normals, mat.transpose(1, 2).inverse())
        if normals_out.shape[0] == 1 and normals.dim() == 2:
            normals_out = normals_out.reshape(normals.shape)
        return normals_out

    def translate(self, *args, **kwargs):
        return self.compose(Translate(device=self.device, *args, **kwargs))

    def scale(self, *args, **kwargs):
        return self.compose(Scale(device=self.device, *args, **kwargs))

    def rotate(self, *args, **kwargs):
        return self.compose(Rotate(device=self.device, *args, **kwargs))

    def rotate_axis_angle(self, *args, **kwargs):
        return self.compose(RotateAxisAngle(device=self.device, *args, **kwargs))

    def clone(self):
        other = Transform3d(dtype=self.dtype, device=self.device)
        if self._lu is not None:
            other._lu = [elem.clone() for elem in self._lu]
        other._matrix = self._matrix.clone()
        other._transforms = [t.clone() for t in self._transforms]
        return other

    def to(self, device, copy: bool = False, dtype=None):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if not copy and (dtype is None or self.dtype == dtype) and self.device == device:
            return self
        other = self.clone()
        other.device = device
        other.dtype = dtype if dtype is not None else other.dtype
        other._matrix = self._matrix.to(device=device, dtype=other.dtype)
        other._transforms = [t.to(device, copy=copy, dtype=dtype) for t in other._transforms]
        return other

    def cpu(self):
        return self.to(torch.device("cpu"))

    def cuda(self):
        return self.to(torch.device("cuda"))


def _coord(c, dtype, device):
    if not torch.is_tensor(c):
        c = torch.tensor(c, dtype=dtype, device=device)
    if c.dim() == 0:
        c = c.view(1)
    return c.to(device=device, dtype=dtype)


def _handle_input(x, y, z, dtype, device, name: str, allow_singleton: bool = False) -> torch.Tensor:
    """x, y, z scalars / (N,) tensors, or x an (N, 3) tensor (or, with allow_singleton, x alone = the same value on
    every axis) -> (N, 3)"""
    if torch.is_tensor(x) and x.dim() == 2:
        if x.shape[1] != 3:
            raise ValueError("Expected tensor of shape (N, 3); got %r (in %s)" % (tuple(x.shape), name))
        if y is not None or z is not None:
            raise ValueError("Expected y and z to be None (in %s)" % name)
        return x.to(device=device, dtype=dtype)
    if not torch.is_tensor(x) and isinstance(x, (list, tuple)) and y is None and z is None and len(x) and \
            isinstance(x[0], (list, tuple)):
        return torch.tensor(x, dtype=dtype, device=device).view(-1, 3)
    if allow_singleton and y is None and z is None:
        y = x
        z = x
    xyz = [_coord(c, dtype, device) for c in (x, y, z)]
    sizes = [c.shape[0] for c in xyz]
    N = max(sizes)
    for c in xyz:
        if c.shape[0] != 1 and c.shape[0] != N:
            raise ValueError("Got non-broadcastable sizes %r (in %s)" % (sizes, name))
    xyz = [c.expand(N) for c in xyz]
    return torch.stack(xyz, dim=1)


class Translate(Transform3d):
    def __init__(self, x, y=None, z=None, dtype: torch.dtype = torch.float32, device="cpu"):
        super().__init__(device=device)
        xyz = _handle_input(x, y, z, dtype, device, "Translate")
        N = xyz.shape[0]
        mat = torch.eye(4, dtype=dtype, device=device)
        mat = mat.view(1, 4, 4).repeat(N, 1, 1)
        mat[:, 3, :3] = xyz
        self._matrix = mat

    def _get_matrix_inverse(self) -> torch.Tensor:
        inv_mask = self._matrix.new_ones([1, 4, 4])
        inv_mask[0, 3, :3] = -1.0
        return self._matrix * inv_mask


class Scale(Transform3d):
    def __init__(self, x, y=None, z=None, dtype: torch.dtype = torch.float32, device="cpu"):
        super().__init__(device=device)
        xyz = _handle_input(x, y, z, dtype, device, "scale", allow_singleton=True)
        N = xyz.shape[0]
        mat = torch.eye(4, dtype=dtype, device=device)
        mat = mat.view(1, 4, 4).repeat(N, 1, 1)
        mat[:, 0, 0] = xyz[:, 0]
        mat[:, 1, 1] = xyz[:, 1]
        mat[:, 2, 2] = xyz[:, 2]
        self._matrix = mat

    def _get_matrix_inverse(self) -> torch.Tensor:
        xyz = torch.stack([self._matrix[:, i, i] for i in range(4)], dim=1)
        ixyz = 1.0 / xyz
        return torch.diag_embed(ixyz, dim1=1, dim2=2)


class Rotate(Transform3d):
    def __init__(self, R: torch.Tensor, dtype: torch.dtype = torch.float32, device="cpu", orthogonal_tol: float = 1e-5):
        super().__init__(device=device)
        if not torch.is_tensor(R):
            R = torch.tensor(R, dtype=dtype)
        if R.dim() == 2:
            R = R[None]
        if R.shape[-2:] != (3, 3):
            raise ValueError("R must have shape (3, 3) or (N, 3, 3); got %s" % repr(R.shape))
        R = R.to(dtype=dtype).to(device=device)
        N = R.shape[0]
        mat = torch.eye(4, dtype=dtype, device=device)
        mat = mat.view(1, 4, 4).repeat(N, 1, 1)
        mat[:, :3, :3] = R
        self._matrix = mat

    def _get_matrix_inverse(self) -> torch.Tensor:
        return self._matrix.permute(0, 2, 1).contiguous()


class RotateAxisAngle(Rotate):
    def __init__(self, angle, axis: str = "X", degrees: bool = True, dtype: torch.dtype = torch.float32, device="cpu"):
        axis = axis.upper()
        if axis not in ("X", "Y", "Z"):
            raise ValueError("Expected axis to be one of ['X', 'Y', 'Z']; got %s" % axis)
        angle = _coord(angle, dtype, device)
        angle = angle * (math.pi / 180.0) if degrees else angle
        cos, sin = torch.cos(angle), torch.sin(angle)
        one, zero = torch.ones_like(angle), torch.zeros_like(angle)
        if axis == "X":
            flat = (one, zero, zero, zero, cos, -sin, zero, sin, cos)
        elif axis == "Y":
            flat = (cos, zero, sin, zero, one, zero, -sin, zero, cos)
        else:
            flat = (cos, -sin, zero, sin, cos, zero, zero, zero, one)
        R = torch.stack(flat, -1).reshape(angle.shape + (3, 3))
        # column-vector rotation matrices act on row vectors through their transpose
        super().__init__(device=device, R=R.transpose(-1, -2), dtype=dtype)
