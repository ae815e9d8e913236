"""PLY vertex / face reader and ascii writer (pytorch3d.io.ply_io interface: load_ply -> (verts, faces))."""
import numpy as np
import torch

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}


def _read(f):
    data = open(f, "rb").read() if isinstance(f, (str, bytes)) or hasattr(f, "__fspath__") else f.read()
    end = data.index(b"end_header")
    end = data.index(b"\n", end) + 1
    header = data[:end].decode("ascii", "replace").splitlines()
    fmt, elements = "ascii", []
    for line in header:
        t = line.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append({"name": t[1], "count": int(t[2]), "props": []})
        elif t[0] == "property":
            elements[-1]["props"].append(t[1:])
    return fmt, elements, data[end:]


def load_ply_arrays(f):
    """-> {element name: {property name: array}} ('vertex_indices' style list properties become (count, 3) arrays)"""
    fmt, elements, body = _read(f)
    out = {}
    if fmt == "ascii":
        tokens = body.decode("ascii", "replace").split()
        pos = 0
        for e in elements:
            cols = {p[-1]: [] for p in e["props"]}
            for _ in range(e["count"]):
                for p in e["props"]:
                    if p[0] == "list":
                        n = int(tokens[pos]); pos += 1
                        cols[p[-1]].append([float(x) for x in tokens[pos:pos + n]]); pos += n
                    else:
                        cols[p[-1]].append(float(tokens[pos])); pos += 1
            out[e["name"]] = {k: np.asarray(v) for k, v in cols.items()}
        return out
    endian = "<" if "little" in fmt else ">"
    off = 0
    for e in elements:
        if all(p[0] != "list" for p in e["props"]):
            dt = np.dtype([(p[-1], endian + _TYPES[p[0]]) for p in e["props"]])
            arr = np.frombuffer(body, dtype=dt, count=e["count"], offset=off)
            off += dt.itemsize * e["count"]
            out[e["name"]] = {n: np.asarray(arr[n]) for n in arr.dtype.names}
        else:
            cols = {p[-1]: [] for p in e["props"]}
            for _ in range(e["count"]):
                for p in e["props"]:
                    if p[0] == "list":
                        ct = np.dtype(endian + _TYPES[p[1]]); it = np.dtype(endian + _TYPES[p[2]])
                        n = int(np.frombuffer(body, ct, 1, off)[0]); off += ct.itemsize
                        cols[p[-1]].append(np.frombuffer(body, it, n, off).tolist()); off += it.itemsize * n
                    else:
                        dt = np.dtype(endian + _TYPES[p[0]])
                        cols[p[-1]].append(np.frombuffer(body, dt, 1, off)[0]); off += dt.itemsize
            out[e["name"]] = {k: np.asarray(v) for k, v in cols.items()}
    return out


def load_ply(f, path_manager=None):
    d = load_ply_arrays(f)
    v = d.get("vertex", {})
    verts = torch.tensor(np.stack([v["x"], v["y"], v["z"]], 1), dtype=torch.float32) if "x" in v else torch.zeros((0, 3))
    faces = torch.zeros((0, 3), dtype=torch.int64)
    face = d.get("face", {})
    for key in ("vertex_indices", "vertex_index"):
        if key in face and len(face[key]):
            faces = torch.tensor(np.asarray(face[key]).reshape(-1, 3), dtype=torch.int64)
    return verts, faces


def save_ply(f, verts, faces=None, verts_normals=None, ascii: bool = True, decimal_places=None, path_manager=None):
    v = verts.detach().cpu().numpy()
    n = verts_normals.detach().cpu().numpy() if verts_normals is not None else None
    fa = faces.detach().cpu().numpy() if faces is not None else np.zeros((0, 3), np.int64)
    opened = isinstance(f, (str, bytes)) or hasattr(f, "__fspath__")
    fh = open(f, "w") if opened else f
    try:
        fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(v))
        if n is not None:
            fh.write("property float nx\nproperty float ny\nproperty float nz\n")
        fh.write("element face %d\nproperty list uchar int vertex_index\nend_header\n" % len(fa))
        for i in range(len(v)):
            row = list(v[i]) + (list(n[i]) if n is not None else [])
            fh.write(" ".join("%g" % x for x in row) + "\n")
        for t in fa:
            fh.write("3 %d %d %d\n" % tuple(int(x) for x in t))
    finally:
        if opened:
            fh.close()
