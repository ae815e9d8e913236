"""Batched point clouds with the interface (public methods AND private attribute names) of
pytorch3d.structures.Pointclouds.  The reference subclasses it (DSS/core/cloud.py:23) and reads / writes the private
tensors directly, so their names and lazy-computation rules follow pytorch3d's documentation.  Independent implementation."""
from typing import List, Optional, Sequence, Tuple, Union

import torch

from . import utils as struct_utils


_device_constant = struct_utils.device_constant


class Pointclouds:
    _INTERNAL_TENSORS = [
        "_points_packed", "_points_padded", "_normals_packed", "_normals_padded", "_features_packed",
        "_features_padded", "_packed_to_cloud_idx", "_cloud_to_packed_first_idx", "_num_points_per_cloud",
        "_padded_to_packed_idx", "valid", "equisized",
    ]

    def __init__(self, points, normals=None, features=None):
        self.device = torch.device("cpu")
        self.equisized = False
        self.valid = None
        self._any_valid = False
        self._sizes_host = []
        self._sizes_src = None      # the size tensor `_sizes_host` was made with
        self._N = 0
        self._P = 0
        self._C = None
        self._points_list = None
        self._normals_list = None
        self._features_list = None
        self._num_points_per_cloud = None
        self._points_packed = None
        self._normals_packed = None
        self._features_packed = None
        self._packed_to_cloud_idx = None
        self._cloud_to_packed_first_idx = None
        self._points_padded = None
        self._normals_padded = None
        self._features_padded = None
        self._padded_to_packed_idx = None

        if isinstance(points, (list, tuple)):
            self._points_list = list(points)
            self._N = len(self._points_list)
            self.valid = torch.zeros((self._N,), dtype=torch.bool, device=self.device)
            if self._N > 0:
                self.device = self._points_list[0].device
                for p in self._points_list:
                    if len(p) > 0 and (p.dim() != 2 or p.shape[1] != 3):
                        raise ValueError("Clouds in list must be of shape Px3 or empty")
                    if p.device != self.device:
                        raise ValueError("All points must be on the same device")
                # the sizes are host integers: nothing here has to ask the device (int(num.max()), num.unique() and the two
                # host-to-device copies each waited for the GPU; train_mvr.py builds ~10 of these objects per iteration)
                sizes = [len(p) for p in self._points_list]
                self._sizes_host = sizes
                self._P = max(sizes)
                self._any_valid = any(n > 0 for n in sizes)
                self.valid = _device_constant(tuple(n > 0 for n in sizes), torch.bool, self.device)
                if len(set(sizes)) == 1:
                    self.equisized = True
                self._num_points_per_cloud = _device_constant(tuple(sizes), torch.int64, self.device)
            else:
                self._num_points_per_cloud = torch.tensor([], dtype=torch.int64)
            self._sizes_src = self._num_points_per_cloud
        elif torch.is_tensor(points):
            if points.dim() != 3 or points.shape[2] != 3:
                raise ValueError("Points tensor has incorrect dimensions.")
            self._points_padded = points
            self._N = points.shape[0]
            self._P = points.shape[1]
            self.device = points.device
            self.valid = torch.ones((self._N,), dtype=torch.bool, device=self.device)
            self._any_valid = self._N > 0
            self._sizes_host = [self._P] * self._N
            self._num_points_per_cloud = torch.full((self._N,), self._P, dtype=torch.int64, device=self.device)
            self._sizes_src = self._num_points_per_cloud
            self.equisized = True
        else:
            raise ValueError("Points must be either a list or a tensor with shape (batch_size, P, 3) where P is the "
                             "maximum number of points in a cloud.")

        normals_parsed = self._parse_auxiliary_input(normals)
        self._normals_list, self._normals_padded, normals_C = normals_parsed
        if normals_C is not None and normals_C != 3:
            raise ValueError("Normals are expected to be 3-dimensional")
        features_parsed = self._parse_auxiliary_input(features)
        self._features_list, self._features_padded, features_C = features_parsed
        if features_C is not None:
            self._C = features_C

    def _parse_auxiliary_input(self, aux_input) -> Tuple[Optional[List[torch.Tensor]], Optional[torch.Tensor], Optional[int]]:
        """normals / features given as a list of (Pi, C) or a padded (N, P, C) tensor -> (list, padded, C)."""
        if aux_input is None or self._N == 0:
            return None, None, None
        aux_input_C = None
        if isinstance(aux_input, (list, tuple)):
            aux_input = list(aux_input)
            if len(aux_input) != self._N:
                raise ValueError("Points and auxiliary input must be the same length.")
            for p, d in zip(self._host_sizes(), aux_input):   # (host integers: iterating the device tensor asks the GPU per cloud)
                if d is None:
                    continue
                if int(p) != d.shape[0]:
                    raise ValueError("A cloud has mismatched numbers of points and inputs")
                if p > 0:
                    if d.dim() != 2:
                        raise ValueError("A cloud auxiliary input must be of shape PxC or empty")
                    if aux_input_C is None:
                        aux_input_C = d.shape[1]
                    if aux_input_C != d.shape[1]:
                        raise ValueError("The clouds must have the same number of channels")
                if d.device != self.device:
                    raise ValueError("All auxiliary inputs must be on the same device as the points.")
            if aux_input_C is None:
                return None, None, None
            return [d if d is not None else torch.zeros((0, aux_input_C), device=self.device) for d in aux_input], None, aux_input_C
        if torch.is_tensor(aux_input):
            if aux_input.dim() != 3:
                raise ValueError("Auxiliary input tensor has incorrect dimensions.")
            if self._N != aux_input.shape[0]:
                raise ValueError("Points and inputs must be the same length.")
            if self._P != aux_input.shape[1]:
                raise ValueError("Inputs tensor must have the right maximum number of points in each cloud.")
            if aux_input.device != self.device:
                raise ValueError("All auxiliary inputs must be on the same device as the points.")
            return None, aux_input, aux_input.shape[2]
        raise ValueError("Auxiliary input must be either a list or a tensor with shape (batch_size, P, C).")

    # ---- container protocol -------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return self._N

    def __getitem__(self, index):
        normals, features = None, None
        if isinstance(index, int):
            idx = [index]
        elif isinstance(index, slice):
            idx = list(range(self._N))[index]
        elif isinstance(index, (list, tuple)):
            idx = list(index)
        elif torch.is_tensor(index):
            if index.dim() != 1 or index.dtype.is_floating_point:
                raise IndexError(index)
            if index.dtype == torch.bool:
                index = index.nonzero().squeeze(1)
            idx = index.tolist()
        else:
            raise IndexError(index)
        pl, nl, fl = self.points_list(), self.normals_list(), self.features_list()
        points = [pl[i] for i in idx]
        if nl is not None:
            normals = [nl[i] for i in idx]
        if fl is not None:
            features = [fl[i] for i in idx]
        return self.__class__(points=points, normals=normals, features=features)

    def _host_sizes(self) -> List[int]:
        """the clouds' sizes as host integers (kept from __init__; an object assembled without it asks the device once)"""
        if len(self._sizes_host) != self._N or self._sizes_src is not self._num_points_per_cloud:
            # (a size tensor put there from outside: ask the device once)
            self._sizes_host = self._num_points_per_cloud.tolist() if torch.is_tensor(self._num_points_per_cloud) else []
            self._sizes_src = self._num_points_per_cloud
            self._any_valid = any(n > 0 for n in self._sizes_host)
        return self._sizes_host

    def isempty(self) -> bool:
        # (pytorch3d asks the device: `self.valid.eq(False).all()`; `valid` is made from host integers in __init__)
        if self._N == 0:
            return True
        self._host_sizes()
        return not self._any_valid

    # ---- list / packed / padded views ---------------------------------------------------------------------------
    def points_list(self) -> List[torch.Tensor]:
        if self._points_list is None:
            assert self._points_padded is not None, "points_padded is required to compute points_list."
            self._points_list = [self._points_padded[i, : int(n)] for i, n in enumerate(self._host_sizes())]
        return self._points_list

    def _aux_list(self, name) -> Optional[List[torch.Tensor]]:
        lst = getattr(self, "_%s_list" % name)
        if lst is None:
            padded = getattr(self, "_%s_padded" % name)
            if padded is None:
                return None
            lst = [padded[i, : int(n)] for i, n in enumerate(self._host_sizes())]
            setattr(self, "_%s_list" % name, lst)
        return lst

    def normals_list(self):
        return self._aux_list("normals")

    def features_list(self):
        return self._aux_list("features")

    def points_packed(self) -> torch.Tensor:
        self._compute_packed()
        return self._points_packed

    def normals_packed(self) -> Optional[torch.Tensor]:
        self._compute_packed()
        return self._normals_packed

    def features_packed(self) -> Optional[torch.Tensor]:
        self._compute_packed()
        return self._features_packed

    def packed_to_cloud_idx(self):
        self._compute_packed()
        return self._packed_to_cloud_idx

    def cloud_to_packed_first_idx(self):
        self._compute_packed()
        return self._cloud_to_packed_first_idx

    def num_points_per_cloud(self) -> torch.Tensor:
        return self._num_points_per_cloud

    def points_padded(self) -> torch.Tensor:
        self._compute_padded()
        return self._points_padded

    def normals_padded(self) -> Optional[torch.Tensor]:
        self._compute_padded()
        return self._normals_padded

    def features_padded(self) -> Optional[torch.Tensor]:
        self._compute_padded()
        return self._features_padded

    def padded_to_packed_idx(self):
        """(sum Pi,) index into the flattened (N * max P) padded tensor of every packed point."""
        if self._padded_to_packed_idx is not None:
            return self._padded_to_packed_idx
        if self._N == 0:
            self._padded_to_packed_idx = []
        else:
            self._padded_to_packed_idx = torch.cat(
                [torch.arange(int(v), dtype=torch.int64, device=self.device) + i * self._P
                 for i, v in enumerate(self._host_sizes())], dim=0)
        return self._padded_to_packed_idx

    def _compute_padded(self, refresh: bool = False):
        if not (refresh or self._points_padded is None):
            return
        self._normals_padded, self._features_padded = None, None
        if self.isempty():
            self._points_padded = torch.zeros((self._N, 0, 3), device=self.device)
            return
        self._points_padded = struct_utils.list_to_padded(self.points_list(), (self._P, 3), pad_value=0.0,
                                                          equisized=self.equisized)
        normals_list = self.normals_list()
        if normals_list is not None:
            self._normals_padded = struct_utils.list_to_padded(normals_list, (self._P, 3), pad_value=0.0,
                                                               equisized=self.equisized)
        features_list = self.features_list()
        if features_list is not None:
            self._features_padded = struct_utils.list_to_padded(features_list, (self._P, self._C), pad_value=0.0,
                                                                equisized=self.equisized)

    def _compute_packed(self, refresh: bool = False):
        if not (refresh or any(v is None for v in [self._points_packed, self._packed_to_cloud_idx,
                                                   self._cloud_to_packed_first_idx])):
            return
        points_list = self.points_list()
        normals_list = self.normals_list()
        features_list = self.features_list()
        if self.isempty():
            self._points_packed = torch.zeros((0, 3), dtype=torch.float32, device=self.device)
            self._packed_to_cloud_idx = torch.zeros((0,), dtype=torch.int64, device=self.device)
            self._cloud_to_packed_first_idx = torch.zeros((0,), dtype=torch.int64, device=self.device)
            self._normals_packed = None
            self._features_packed = None
            return
        packed, num, first, to_cloud = struct_utils.list_to_packed(points_list)
        self._points_packed = packed
        self._packed_to_cloud_idx = to_cloud
        self._cloud_to_packed_first_idx = first
        self._normals_packed, self._features_packed = None, None
        if normals_list is not None:
            self._normals_packed = torch.cat(normals_list, 0)     # (the index tensors are those of the points)
        if features_list is not None:
            self._features_packed = torch.cat(features_list, 0)

    # ---- copies / devices ---------------------------------------------------------------------------------------
    def _copy_with(self, fn):
        other = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                setattr(other, k, fn(v))
            elif isinstance(v, list) and v and all(torch.is_tensor(t) for t in v):
                setattr(other, k, [fn(t) for t in v])
            elif hasattr(v, "clone") and not isinstance(v, (int, float, bool, str)) and not torch.is_tensor(v):
                try:
                    setattr(other, k, v.clone())
                except Exception:  # noqa: BLE001
                    setattr(other, k, v)
            else:
                setattr(other, k, v)
        return other

    def clone(self):
        return self._copy_with(lambda t: t.clone())

    def detach(self):
        return self._copy_with(lambda t: t.detach())

    def to(self, device, copy: bool = False):
        device = torch.device(device) if not isinstance(device, torch.device) else device
        if not copy and self.device == device:
            return self
        other = self.clone()
        if self.device == device:
            return other
        other.device = device
        for k, v in list(other.__dict__.items()):
            if torch.is_tensor(v):
                setattr(other, k, v.to(device))
            elif isinstance(v, list) and v and all(torch.is_tensor(t) for t in v):
                setattr(other, k, [t.to(device) for t in v])
            elif hasattr(v, "to") and not isinstance(v, (int, float, bool, str)) and k not in ("device",):
                try:
                    setattr(other, k, v.to(device))
                except Exception:  # noqa: BLE001
                    pass
        return other

    def cpu(self):
        return self.to(torch.device("cpu"))

    def cuda(self):
        return self.to(torch.device("cuda"))

    def get_cloud(self, index: int):
        if not isinstance(index, int):
            raise ValueError("Cloud index must be an integer.")
        if index < 0 or index > self._N:
            raise ValueError("Cloud index must be in the range [0, N) where N is the number of clouds in the batch.")
        points = self.points_list()[index]
        normals, features = None, None
        if self.normals_list() is not None:
            normals = self.normals_list()[index]
        if self.features_list() is not None:
            features = self.features_list()[index]
        return points, normals, features

    def split(self, split_sizes: list):
        if not all(isinstance(x, int) for x in split_sizes):
            raise ValueError("Value of split_sizes must be a list of integers.")
        out, k = [], 0
        for s in split_sizes:
            out.append(self[k:k + s])
            k += s
        return out

    # ---- in-place geometry --------------------------------------------------------------------------------------
    def offset_(self, offsets_packed):
        points_packed = self.points_packed()
        if offsets_packed.shape != points_packed.shape:
            raise ValueError("Offsets must have dimension (all_p, 3).")
        self._points_packed = points_packed + offsets_packed
        new_points_list = list(self._points_packed.split(list(self._host_sizes()), 0))
        self._points_list = new_points_list
        if self._points_padded is not None:
            for i, points in enumerate(new_points_list):
                if len(points) > 0:
                    self._points_padded[i, : points.shape[0], :] = points
        return self

    def offset(self, offsets_packed):
        return self.clone().offset_(offsets_packed)

    def scale_(self, scale):
        if not torch.is_tensor(scale):
            scale = torch.full((len(self),), scale, device=self.device)
        new_points_list = []
        points_list = self.points_list()
        for i, old_points in enumerate(points_list):
            new_points_list.append(scale[i] * old_points)
        self._points_list = new_points_list
        if self._points_packed is not None:
            self._points_packed = torch.cat(new_points_list, dim=0)
        if self._points_padded is not None:
            for i, points in enumerate(new_points_list):
                if len(points) > 0:
                    self._points_padded[i, : points.shape[0], :] = points
        return self

    def scale(self, scale):
        return self.clone().scale_(scale)

    def get_bounding_boxes(self) -> torch.Tensor:
        """(N, 3, 2): min and max along every axis."""
        all_mins, all_maxes = [], []
        for points in self.points_list():
            all_mins.append(points.min(dim=0)[0])
            all_maxes.append(points.max(dim=0)[0])
        return torch.stack([torch.stack(all_mins, 0), torch.stack(all_maxes, 0)], dim=2)

    def estimate_normals(self, neighborhood_size: int = 50, disambiguate_directions: bool = True,
                         assign_to_self: bool = False):
        from ..ops import estimate_pointcloud_normals
        normals_est = estimate_pointcloud_normals(self, neighborhood_size=neighborhood_size,
                                                  disambiguate_directions=disambiguate_directions)
        if assign_to_self:
            _, self._normals_padded, _ = self._parse_auxiliary_input(normals_est)
            self._normals_list, self._normals_packed = None, None
            if self._points_list is not None:
                self.normals_list()
            if self._points_packed is not None:
                self._normals_packed = torch.cat(self._normals_list, dim=0)
        return normals_est

    def extend(self, N: int):
        if not isinstance(N, int):
            raise ValueError("N must be an integer.")
        if N <= 0:
            raise ValueError("N must be > 0.")
        new_points_list, new_normals_list, new_features_list = [], None, None
        for points in self.points_list():
            new_points_list.extend(points.clone() for _ in range(N))
        if self.normals_list() is not None:
            new_normals_list = []
            for normals in self.normals_list():
                new_normals_list.extend(normals.clone() for _ in range(N))
        if self.features_list() is not None:
            new_features_list = []
            for features in self.features_list():
                new_features_list.extend(features.clone() for _ in range(N))
        return self.__class__(points=new_points_list, normals=new_normals_list, features=new_features_list)

    def update_padded(self, new_points_padded, new_normals_padded=None, new_features_padded=None):
        def check_shapes(x, size):
            if x.shape[0] != size[0]:
                raise ValueError("new values must have the same batch dimension.")
            if x.shape[1] != size[1]:
                raise ValueError("new values must have the same number of points.")
            if size[2] is not None and x.shape[2] != size[2]:
                raise ValueError("new values must have the same dimension.")

        check_shapes(new_points_padded, [self._N, self._P, 3])
        if new_normals_padded is not None:
            check_shapes(new_normals_padded, [self._N, self._P, 3])
        if new_features_padded is not None:
            check_shapes(new_features_padded, [self._N, self._P, self._C])
        new = self.__class__(points=new_points_padded, normals=new_normals_padded, features=new_features_padded)
        new.equisized = self.equisized
        if new_normals_padded is None:
            new._normals_list = self._normals_list
            new._normals_padded = self._normals_padded
            new._normals_packed = self._normals_packed
        if new_features_padded is None:
            new._features_list = self._features_list
            new._features_padded = self._features_padded
            new._features_packed = self._features_packed
            new._C = self._C
        for k in ("_num_points_per_cloud", "_cloud_to_packed_first_idx", "_packed_to_cloud_idx", "_padded_to_packed_idx", "valid"):
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(new, k, v)
        new._sizes_host, new._sizes_src, new._any_valid = list(self._host_sizes()), new._num_points_per_cloud, self._any_valid
        if new._padded_to_packed_idx is None and self._N > 0:
            new.padded_to_packed_idx()
        if self._N > 0 and torch.is_tensor(new.padded_to_packed_idx()):
            idx = new.padded_to_packed_idx()
            new._points_packed = new_points_padded.reshape(-1, 3)[idx]
            new._points_list = None
            new.points_list()
            if new_normals_padded is not None:
                new._normals_packed = new_normals_padded.reshape(-1, 3)[idx]
            if new_features_padded is not None:
                new._features_packed = new_features_padded.reshape(-1, new_features_padded.shape[-1])[idx]
            if new._cloud_to_packed_first_idx is None:
                new._points_packed = None  # let _compute_packed rebuild everything consistently
        return new

    def inside_box(self, box):
        if box.dim() > 3 or box.dim() < 2:
            raise ValueError("Input box must be of shape (2, 3) or (N, 2, 3).")
        if box.dim() == 3 and box.shape[0] != 1 and box.shape[0] != self._N:
            raise ValueError("Input box dimension is incompatible with pointcloud size.")
        if box.dim() == 2:
            box = box[None]
        if (box[..., 0, :] > box[..., 1, :]).any():
            raise ValueError("Input box is invalid: min values larger than max values.")
        points_packed = self.points_packed()
        sumP = points_packed.shape[0]
        if box.shape[0] == 1:
            box = box.expand(sumP, 2, 3)
        elif box.shape[0] == self._N:
            box = box.unbind(0)
            box = [b.expand(int(p), 2, 3) for (b, p) in zip(box, self._host_sizes())]
            box = torch.cat(box, 0)
        coord_inside = (points_packed >= box[:, 0]) * (points_packed <= box[:, 1])
        return coord_inside.all(dim=-1)
