"""Minimal triangle-mesh batch with the pytorch3d.structures.Meshes method names the reference touches
(config.py:116-121: `ico_sphere(...).scale_verts_`, `sample_points_from_meshes`; dataset.py / visualize.py:
`verts_packed`, `faces_packed`, `verts_list`, `faces_list`)."""
from typing import List

import torch

from . import utils as struct_utils


class Meshes:
    def __init__(self, verts=None, faces=None, textures=None):
        if torch.is_tensor(verts):
            verts = list(verts.unbind(0))
        if torch.is_tensor(faces):
            faces = [f[(f >= 0).all(-1)] for f in faces.unbind(0)]
        self._verts_list: List[torch.Tensor] = list(verts or [])
        self._faces_list: List[torch.Tensor] = [f.to(torch.int64) for f in (faces or [])]
        if len(self._verts_list) != len(self._faces_list):
            raise ValueError("verts and faces must have the same batch size")
        self.textures = textures
        self.device = self._verts_list[0].device if self._verts_list else torch.device("cpu")
        self._N = len(self._verts_list)

    def __len__(self):
        return self._N

    def __getitem__(self, index):
        if isinstance(index, int):
            index = [index]
        elif isinstance(index, slice):
            index = list(range(self._N))[index]
        return Meshes([self._verts_list[i] for i in index], [self._faces_list[i] for i in index])

    def isempty(self) -> bool:
        return self._N == 0 or all(v.shape[0] == 0 for v in self._verts_list)

    def verts_list(self):
        return self._verts_list

    def faces_list(self):
        return self._faces_list

    def num_verts_per_mesh(self):
        return torch.tensor([v.shape[0] for v in self._verts_list], dtype=torch.int64, device=self.device)

    def num_faces_per_mesh(self):
        return torch.tensor([f.shape[0] for f in self._faces_list], dtype=torch.int64, device=self.device)

    def mesh_to_verts_packed_first_idx(self):
        n = self.num_verts_per_mesh()
        return torch.cumsum(n, 0) - n

    def mesh_to_faces_packed_first_idx(self):
        n = self.num_faces_per_mesh()
        return torch.cumsum(n, 0) - n

    def verts_packed(self):
        return torch.cat(self._verts_list, 0) if self._N else torch.zeros((0, 3))

    def faces_packed(self):
        """faces with vertex indices offset into verts_packed"""
        if not self._N:
            return torch.zeros((0, 3), dtype=torch.int64)
        first = self.mesh_to_verts_packed_first_idx()
        return torch.cat([f + first[i] for i, f in enumerate(self._faces_list)], 0)

    def faces_packed_to_mesh_idx(self):
        return torch.repeat_interleave(torch.arange(self._N, device=self.device), self.num_faces_per_mesh())

    def verts_packed_to_mesh_idx(self):
        return torch.repeat_interleave(torch.arange(self._N, device=self.device), self.num_verts_per_mesh())

    def verts_padded(self):
        return struct_utils.list_to_padded(self._verts_list, pad_value=0.0)

    def faces_padded(self):
        return struct_utils.list_to_padded(self._faces_list, pad_value=-1)

    def faces_areas_packed(self):
        v, f = self.verts_packed(), self.faces_packed()
        return 0.5 * torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1).norm(dim=1)

    def faces_normals_packed(self):
        v, f = self.verts_packed(), self.faces_packed()
        n = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1)
        return torch.nn.functional.normalize(n, dim=1, eps=1e-6)

    def verts_normals_packed(self):
        """area-weighted average of the adjacent face normals"""
        v, f = self.verts_packed(), self.faces_packed()
        fn = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1)
        vn = torch.zeros_like(v)
        for k in range(3):
            vn = vn.index_add(0, f[:, k], fn)
        return torch.nn.functional.normalize(vn, dim=1, eps=1e-6)

    def verts_normals_list(self):
        return list(self.verts_normals_packed().split(self.num_verts_per_mesh().tolist(), 0))

    def scale_verts_(self, scale):
        if not torch.is_tensor(scale):
            scale = torch.full((self._N,), float(scale), device=self.device)
        self._verts_list = [v * scale[i] for i, v in enumerate(self._verts_list)]
        return self

    def scale_verts(self, scale):
        return self.clone().scale_verts_(scale)

    def offset_verts_(self, vert_offsets_packed):
        if vert_offsets_packed.dim() == 1:
            vert_offsets_packed = vert_offsets_packed[None].expand(int(self.num_verts_per_mesh().sum()), 3)
        off = vert_offsets_packed.split(self.num_verts_per_mesh().tolist(), 0)
        self._verts_list = [v + o for v, o in zip(self._verts_list, off)]
        return self

    def offset_verts(self, vert_offsets_packed):
        return self.clone().offset_verts_(vert_offsets_packed)

    def get_bounding_boxes(self):
        mins = torch.stack([v.min(0)[0] for v in self._verts_list])
        maxs = torch.stack([v.max(0)[0] for v in self._verts_list])
        return torch.stack([mins, maxs], dim=2)

    def clone(self):
        return Meshes([v.clone() for v in self._verts_list], [f.clone() for f in self._faces_list], self.textures)

    def detach(self):
        return Meshes([v.detach() for v in self._verts_list], [f.detach() for f in self._faces_list], self.textures)

    def to(self, device, copy: bool = False):
        return Meshes([v.to(device) for v in self._verts_list], [f.to(device) for f in self._faces_list], self.textures)

    def cpu(self):
        return self.to("cpu")

    def cuda(self):
        return self.to("cuda")

    def extend(self, N: int):
        vs, fs = [], []
        for v, f in zip(self._verts_list, self._faces_list):
            vs.extend(v.clone() for _ in range(N))
            fs.extend(f.clone() for _ in range(N))
        return Meshes(vs, fs)


def join_meshes_as_batch(meshes, include_textures: bool = True):
    vs, fs = [], []
    for m in meshes:
        vs.extend(m.verts_list())
        fs.extend(m.faces_list())
    return Meshes(vs, fs)
