"""Wavefront OBJ geometry reader / writer (pytorch3d.io.obj_io interface; vertices + triangulated faces only)."""
from collections import namedtuple

import torch

from ..structures import Meshes

_Faces = namedtuple("Faces", "verts_idx normals_idx textures_idx materials_idx")
_Aux = namedtuple("Properties", "normals verts_uvs material_colors texture_images texture_atlas")


def load_obj(f, load_textures: bool = True, create_texture_atlas: bool = False, texture_atlas_size: int = 4,
             texture_wrap="repeat", device="cpu", path_manager=None):
    verts, normals, faces = [], [], []
    opened = isinstance(f, (str, bytes)) or hasattr(f, "__fspath__")
    fh = open(f, "r") if opened else f
    try:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(x) for x in t[1:4]])
            elif t[0] == "vn":
                normals.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                ids = []
                for tok in t[1:]:
                    i = int(tok.split("/")[0])
                    ids.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(ids) - 1):  # fan triangulation
                    faces.append([ids[0], ids[k], ids[k + 1]])
    finally:
        if opened:
            fh.close()
    v = torch.tensor(verts, dtype=torch.float32, device=device).reshape(-1, 3)
    fi = torch.tensor(faces, dtype=torch.int64, device=device).reshape(-1, 3)
    n = torch.tensor(normals, dtype=torch.float32, device=device).reshape(-1, 3) if normals else None
    neg = torch.full_like(fi, -1)
    return v, _Faces(fi, neg, neg, torch.full((fi.shape[0],), -1, dtype=torch.int64, device=device)), \
        _Aux(n, None, None, None, None)


def load_objs_as_meshes(files, device=None, load_textures: bool = True, create_texture_atlas: bool = False,
                        texture_atlas_size: int = 4, texture_wrap="repeat", path_manager=None):
    vs, fs = [], []
    for f in files:
        v, faces, _ = load_obj(f, device=device or "cpu")
        vs.append(v)
        fs.append(faces.verts_idx)
    return Meshes(vs, fs)


def save_obj(f, verts, faces=None, decimal_places=None, **kwargs):
    fmt = "%f" if decimal_places is None else "%%.%df" % decimal_places
    opened = isinstance(f, (str, bytes)) or hasattr(f, "__fspath__")
    fh = open(f, "w") if opened else f
    try:
        for v in verts.detach().cpu().tolist():
            fh.write("v " + " ".join(fmt % x for x in v) + "\n")
        if faces is not None:
            for t in (faces.detach().cpu() + 1).tolist():
                fh.write("f %d %d %d\n" % tuple(t))
    finally:
        if opened:
            fh.close()
