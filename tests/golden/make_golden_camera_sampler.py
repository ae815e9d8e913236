"""Generates tests/golden/ref_camera_sampler.npz: the parameters the REFERENCE's `CameraSampler`
(/root/reference/DSS/core/camera.py:6-73) draws for a fixed torch seed, captured by replacing the (absent)
pytorch3d `look_at_view_transform` it imports with a recorder.  Pins the draw order and the arithmetic of the mirror
`dss_amd.cameras.CameraSampler`.

    python tests/golden/make_golden_camera_sampler.py
"""
import importlib
import os

import numpy as np
import torch

import make_golden_setup as base  # noqa: F401  (stubs + /root/reference on sys.path)

HERE = os.path.dirname(os.path.abspath(__file__))
ref_camera = importlib.import_module("DSS.core.camera")  # the UNMODIFIED reference module
seen = {}


def recorder(dist, elev, azim, at=None, degrees=True, **kw):
    seen.update(dist=dist.clone(), elev=elev.clone(), azim=azim.clone(), at=at.clone(), degrees=degrees)
    n = dist.shape[0]
    return torch.eye(3)[None].expand(n, 3, 3), torch.zeros(n, 3)


ref_camera.look_at_view_transform = recorder


class _Cams:
    def __init__(self, R=None, T=None, **kw):
        self.R, self.T, self.kw = R, T, kw


def main():
    out = {}
    for tag, (seed, total, batch, rng, sort) in {"a": (0, 128, 8, [[1.2, 2.2]], True), "b": (7, 10, 4, [[5.0, 10.0]], False)}.items():
        torch.manual_seed(seed)
        sampler = ref_camera.CameraSampler(total, batch, distance_range=torch.tensor(rng), sort_distance=sort,
                                           camera_type=_Cams, camera_params={"znear": 0.1})
        assert seen["degrees"] is True
        batches = [c.R.shape[0] for c in sampler]
        out[tag + "_args"] = np.array([seed, total, batch, rng[0][0], rng[0][1], int(sort)], np.float64)
        for k in ("dist", "elev", "azim", "at"):
            out[tag + "_" + k] = seen[k].numpy()
        out[tag + "_batches"] = np.array(batches)
    path = os.path.join(HERE, "ref_camera_sampler.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
