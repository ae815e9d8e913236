"""Minimal packed/padded point-cloud container with the accessor names of pytorch3d's
``Pointclouds`` / DSS's ``PointClouds3D`` (DSS/core/cloud.py) that the splatting hot path calls
(rasterizer.py:155, 194, 236-240, 275-280, 651; renderer.py:57).  pytorch3d itself is absent here
[third party]; any object exposing the same accessors can be passed to ``dss_amd`` instead.
"""
from typing import List, Optional, Sequence, Union

import torch


def _to_list(x) -> Optional[List[torch.Tensor]]:
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        if x.dim() == 2:
            return [x]
        return [x[i] for i in range(x.shape[0])]
    return list(x)


class PointClouds3D:
    def __init__(self, points, normals=None, features=None):
        self._points = _to_list(points)
        self._normals = _to_list(normals)
        self._features = _to_list(features)
        self.device = self._points[0].device if self._points else torch.device("cpu")
        self._num = torch.tensor([p.shape[0] for p in self._points], dtype=torch.int64, device=self.device)
        self._first = torch.cumsum(self._num, 0) - self._num
        self._packed_cache = {}

    def __len__(self):
        return len(self._points)

    def isempty(self) -> bool:
        return len(self._points) == 0 or int(sum(p.shape[0] for p in self._points)) == 0

    def num_points_per_cloud(self):
        return self._num

    def cloud_to_packed_first_idx(self):
        return self._first

    def packed_to_cloud_idx(self):
        return torch.repeat_interleave(torch.arange(len(self), device=self.device), self._num)

    def _packed(self, name):
        lst = getattr(self, "_" + name)
        if lst is None:
            return None
        if name not in self._packed_cache:
            self._packed_cache[name] = lst[0] if len(lst) == 1 else torch.cat(lst, dim=0)
        return self._packed_cache[name]

    def points_list(self):
        return self._points

    def normals_list(self):
        return self._normals

    def features_list(self):
        return self._features

    def points_packed(self):
        return self._packed("points")

    def normals_packed(self):
        return self._packed("normals")

    def features_packed(self):
        return self._packed("features")

    def _padded(self, lst):
        if lst is None:
            return None
        mx = int(max(t.shape[0] for t in lst))
        if all(t.shape[0] == mx for t in lst):
            return torch.stack(lst, 0)
        out = lst[0].new_zeros((len(lst), mx) + tuple(lst[0].shape[1:]))
        for i, t in enumerate(lst):
            out[i, : t.shape[0]] = t
        return out

    def points_padded(self):
        return self._padded(self._points)

    def normals_padded(self):
        return self._padded(self._normals)

    def features_padded(self):
        return self._padded(self._features)

    def extend(self, N: int) -> "PointClouds3D":
        """N copies of every cloud (shared autograd graph), like Pointclouds.extend."""
        rep = lambda lst: None if lst is None else [t for t in lst for _ in range(N)]
        return PointClouds3D(rep(self._points), rep(self._normals), rep(self._features))

    def to(self, device):
        mv = lambda lst: None if lst is None else [t.to(device) for t in lst]
        return PointClouds3D(mv(self._points), mv(self._normals), mv(self._features))
