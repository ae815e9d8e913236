"""CPU: host-side logic of the drop-in layer that needs no GPU (camera matrix cache, packed masks)."""
import numpy as np
import pytest
import torch

import scenes
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
from dss_amd.cloud import PointClouds3D
from dss_amd.losses import _packed_mask


def test_camera_matrices_follow_the_pytorch3d_convention_and_scenes_helper():
    R, T = look_at_view_transform(2.0, 30.0, [45.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T)
    M, V, _ = scenes.camera_matrices(2.0, 30.0, 45.0)
    assert np.allclose(cams.get_full_projection_transform().get_matrix().numpy(), M, atol=1e-6)
    assert np.allclose(cams.get_world_to_view_transform().get_matrix().numpy(), V, atol=1e-6)


def test_camera_matrix_cache_is_invalidated_by_any_change():
    R, T = look_at_view_transform(2.0, 30.0, [45.0, 90.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T)
    m1 = cams.get_full_projection_transform().get_matrix()
    assert cams.get_full_projection_transform().get_matrix() is m1            # served from the cache
    cams.T.add_(0.1)                                                           # in-place edit: version counter
    m2 = cams.get_full_projection_transform().get_matrix()
    assert not torch.equal(m1, m2)
    cams.R = look_at_view_transform(2.5, 10.0, [45.0, 90.0])[0]                # replaced tensor: identity
    m3 = cams.get_full_projection_transform().get_matrix()
    assert not torch.equal(m2, m3)
    cams.znear = torch.full((2,), 0.5)
    assert not torch.equal(cams.get_full_projection_transform().get_matrix(), m3)
    # explicit overrides bypass the cache and do not poison it
    R2, T2 = look_at_view_transform(3.0, 0.0, [0.0, 10.0])
    over = cams.get_full_projection_transform(R=R2, T=T2).get_matrix()
    again = cams.get_full_projection_transform().get_matrix()
    assert not torch.equal(over, again)
    fresh = FoVPerspectiveCameras(znear=0.5, R=cams.R, T=cams.T)
    assert torch.allclose(fresh.get_full_projection_transform().get_matrix(), again)
    moved = cams.to("cpu")
    assert torch.equal(moved.get_full_projection_transform().get_matrix(), again)


def test_packed_mask_accepts_padded_per_camera_and_packed_layouts():
    a, b = torch.rand(5, 3), torch.rand(3, 3)
    pc = PointClouds3D([a, b])
    padded = torch.tensor([[1, 0, 1, 1, 0], [0, 1, 1, 0, 0]], dtype=torch.bool)
    assert _packed_mask(padded, pc).tolist() == [True, False, True, True, False, False, True, True]
    assert _packed_mask(torch.ones(8, dtype=torch.bool), pc).all()
    one = PointClouds3D([a])
    per_camera = torch.tensor([[1, 0, 0, 0, 0], [0, 0, 0, 0, 1]], dtype=torch.bool)   # (cameras, P): OR over the rows
    assert _packed_mask(per_camera, one).tolist() == [True, False, False, False, True]
    with pytest.raises(ValueError):
        _packed_mask(torch.ones(3, 5, dtype=torch.bool), pc)
    with pytest.raises(ValueError):
        _packed_mask(torch.ones(7, dtype=torch.bool), pc)
    assert _packed_mask(None, pc) is None
