"""Unit icosphere (pytorch3d.utils.ico_sphere interface): an icosahedron subdivided `level` times, every new vertex
pushed onto the unit sphere.  level 4 -> 2562 vertices, 5120 faces."""
import math

import torch

from ..structures import Meshes


def _icosahedron():
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    verts = torch.tensor(v, dtype=torch.float32)
    return verts / verts.norm(dim=1, keepdim=True), torch.tensor(f, dtype=torch.int64)


def _subdivide(verts, faces):
    edges = torch.cat([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    edges = torch.sort(edges, dim=1)[0]
    uniq, inverse = torch.unique(edges, dim=0, return_inverse=True)
    mid = verts[uniq].mean(1)
    mid = mid / mid.norm(dim=1, keepdim=True)
    V, F_ = verts.shape[0], faces.shape[0]
    m01, m12, m20 = inverse[:F_] + V, inverse[F_:2 * F_] + V, inverse[2 * F_:] + V
    f0, f1, f2 = faces[:, 0], faces[:, 1], faces[:, 2]
    new_faces = torch.cat([torch.stack([f0, m01, m20], 1), torch.stack([f1, m12, m01], 1), torch.stack([f2, m20, m12], 1),
                           torch.stack([m01, m12, m20], 1)], 0)
    return torch.cat([verts, mid], 0), new_faces


def ico_sphere(level: int = 0, device=None):
    if level < 0:
        raise ValueError("level must be >= 0.")
    verts, faces = _icosahedron()
    for _ in range(level):
        verts, faces = _subdivide(verts, faces)
    if device is not None:
        verts, faces = verts.to(device), faces.to(device)
    return Meshes(verts=[verts], faces=[faces])
