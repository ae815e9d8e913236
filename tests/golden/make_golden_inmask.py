"""Generates tests/golden/ref_inmask.npz: the in-mask filter of the reference model
(/root/reference/DSS/models/point_modeling.py:183-208) = the reference's own `get_tensor_values`
(/root/reference/DSS/utils/__init__.py:266-317, i.e. torch's F.grid_sample bilinear / reflection) evaluated at
the projections of the points, `.bool()`, any over the views, & visibility.  The projection uses dss_amd's camera
stand-in (pytorch3d convention, pinned elsewhere); everything runs on CPU torch.

    python tests/golden/make_golden_inmask.py
"""
import importlib
import os

import numpy as np
import torch

import make_golden_setup as base  # noqa: F401  (stubs + /root/reference on sys.path)
import scenes
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform

HERE = os.path.dirname(os.path.abspath(__file__))
ref_utils = importlib.import_module("DSS.utils")  # the UNMODIFIED reference module


def main():
    rng = np.random.default_rng(9)
    pts, _ = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts).astype(np.float32)
    P = pts.shape[0]
    H, W = 96, 128
    R, T = look_at_view_transform(2.0, 25.0, [20.0, 140.0, 260.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T)
    N = 3
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.stack([(((yy - H * 0.5) / (0.35 * H)) ** 2 + ((xx - W * (0.4 + 0.1 * n)) / (0.3 * W)) ** 2 < 1.0)
                     for n in range(N)]).astype(np.float32)
    visibility = rng.random(P) < 0.8
    points = torch.from_numpy(pts)[None].expand(N, -1, -1)
    p_screen = cams.transform_points(points)                       # (N,P,3), like self.cameras.transform_points
    p = -p_screen[..., :2]
    # a few raw sampling positions including the borders, for the grid_sample restatement on its own
    probe = torch.from_numpy(rng.uniform(-1, 1, (N, 4000, 2)).astype(np.float32))
    probe[:, :8] = torch.tensor([[-1, -1], [1, 1], [-1, 1], [1, -1], [0, 0], [0.999, -0.999], [-1, 0.3], [0.3, 1]])
    t_mask = torch.from_numpy(mask)[:, None]
    probe_vals = ref_utils.get_tensor_values(t_mask, probe, squeeze_channel_dim=True)
    soft = torch.from_numpy(rng.random((N, 1, H, W)).astype(np.float32))     # non-binary image: checks the weights
    probe_soft = ref_utils.get_tensor_values(soft, probe, squeeze_channel_dim=True)
    mask_pred = ref_utils.get_tensor_values(t_mask, p.clamp(-1.0, 1.0), squeeze_channel_dim=True).bool()
    inmask = mask_pred.any(dim=0, keepdim=True) & torch.from_numpy(visibility)[None]
    out = {"points": pts, "M": cams.get_full_projection_transform().get_matrix().numpy(), "mask": mask,
           "visibility": visibility, "inmask": inmask[0].numpy(), "per_view": mask_pred.numpy(),
           "probe": probe.numpy(), "probe_vals": probe_vals.numpy(), "soft": soft[:, 0].numpy(),
           "probe_soft": probe_soft.numpy()}
    path = os.path.join(HERE, "ref_inmask.npz")
    np.savez_compressed(path, **out)
    print("in mask:", int(inmask.sum()), "of", P, "| wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
