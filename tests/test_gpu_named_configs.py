"""GPU: parity at the sizes BASELINE.json names (configs[2..4]) -- the whole fused hot path through the C ABI
(dss_render_forward = setup + binning + fine + blend; dss_render_backward = blend backward + median radius + occupancy
backward + clip) against the oracle on the same seeded inputs:

  cfg3  yoga6_out.ply x10 tangent-plane jitter = 99,790 points, 8 cameras of the reference's CameraSampler rule
        (DSS/core/camera.py:41-51; every 16th of the 128 distance-sorted views, 2.2 -> 1.2), 512x512
  cfg4  synthetic 1M-point cloud (SURVEY 8d generator), all 8 ring cameras, 1024x1024
  cfg5  synthetic 4M-point cloud, 1 camera, 2048x2048

Bar: per-point screen records and fragments (idx, zbuf, qvalue, occupancy, visibility, search radius) bit-exact, RGBA
<= 1e-4, gradients rel-L2 <= 1e-3.  The oracle's windowed forward and point-centric backward are O(points x window) and
run multi-threaded on the host (oracle/dss_oracle.c), so each case stays within about two minutes.
"""
import time

import numpy as np
import pytest
import torch

import oracle
import scenes
from dss_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K, THR, RADII_S, CLIP, CUTOFF, SIGMA = 5, 0.05, 5.0, 0.05, 1.0, 1.0


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _cfg3():
    from dss_amd.cameras import CameraSampler, FoVPerspectiveCameras
    pts, nrm = scenes.load_cloud("yoga6")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, 10, seed=0)
    assert pts.shape[0] == 99_790
    col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
    torch.manual_seed(0)
    sampler = CameraSampler(128, 8, distance_range=[[1.2, 2.2]], sort_distance=True)
    cam = FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=sampler.R[::16], T=sampler.T[::16])  # far to near
    M = cam.get_full_projection_transform().get_matrix().numpy().astype(np.float32)
    V = cam.get_world_to_view_transform().get_matrix().numpy().astype(np.float32)
    return pts, nrm, col, M, V, 512, scenes.global_h(pts)


def _synthetic(P, S, n_cams):
    """the scene `bench.py --workload cfg4|cfg5` times: same generator, same cameras, same variance scale (scenes.large_cloud_h)"""
    pts, nrm, col = scenes.synthetic_cloud(P, seed=0)
    M, V, _ = scenes.camera_matrices(*scenes.BENCH_CAMERA, scenes.bench_azimuths(n_cams))
    return pts, nrm, col, M, V, S, scenes.large_cloud_h(pts)


CONFIGS = {
    "cfg3": _cfg3,
    "cfg4": lambda: _synthetic(1_000_000, 1024, 8),
    "cfg5": lambda: _synthetic(4_000_000, 2048, 1),
}


@pytest.mark.parametrize("name", ["cfg3", "cfg4", "cfg5"])
def test_named_config_forward_backward_vs_oracle(name):
    t0 = time.time()
    pts, nrm, col, M, V, S, h = CONFIGS[name]()
    N, Pc = M.shape[0], pts.shape[0]
    P = N * Pc
    # ---- oracle (CPU): per-point setup, windowed forward, blend, backward --------------------------------
    sc = scenes.setup_scene(pts, nrm, M, V, S, cutoff=CUTOFF, sigma=SIGMA, h=h, colors=col)
    assert np.array_equal(sc["num_pts"], np.full(N, Pc)), "the scene is meant to have nothing culled"
    o_idx, o_z, o_q, o_occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"],
                                                  sc["first_idx"], sc["num_pts"], S, K, THR)
    o_img = oracle.blend_forward(o_idx, o_q, o_occ, sc["scaler"], sc["colors"])
    grad_out = np.random.default_rng(1).standard_normal((N, S, S, 4)).astype(np.float32)
    o_gf, o_gocc = oracle.blend_backward(grad_out, o_idx, o_q, sc["scaler"], P)
    o_gp, o_vis, o_rs = oracle.splat_backward(sc["points"], sc["radii"], o_idx, o_gocc, None, sc["first_idx"],
                                              sc["num_pts"], RADII_S, CLIP)
    t_oracle = time.time() - t0
    # ---- HIP (fused path, exactly what bench.py times) ---------------------------------------------------
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV, dtype=torch.int64) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    f = ops.render_forward(t(pts), t(nrm), torch.full((N,), h, device=DEV), t(M), t(V), torch.full((N,), 0.1, device=DEV),
                           torch.full((N,), 100.0, device=DEV), first, num, t(np.tile(col, (N, 1))), S, K, CUTOFF, THR,
                           SIGMA, False, True)
    for key, want in (("pts_screen", sc["points"]), ("ellipse_params", sc["ellipse"]), ("radii", sc["radii"]),
                      ("scaler", sc["scaler"]), ("cutoff_threshold", sc["cutoff"])):
        assert np.array_equal(f[key].cpu().numpy(), want), key
    assert bool(f["valid"].all())
    assert np.array_equal(f["idx"].cpu().numpy(), o_idx)
    assert np.array_equal(f["zbuf"].cpu().numpy(), o_z)
    assert np.array_equal(f["qvalue"].cpu().numpy(), o_q)
    assert np.array_equal(f["occupancy"].cpu().numpy(), o_occ)
    assert np.array_equal(f["visible"].cpu().numpy(), o_vis)
    err = float(np.abs(f["image"].cpu().numpy() - o_img).max())
    assert err <= 1e-4, err
    g_feat, g_pts, rs = ops.render_backward(t(grad_out), f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"],
                                            f["radii"], f["visible"], first, num, RADII_S, CLIP, return_rs=True)
    assert np.array_equal(rs.cpu().numpy(), o_rs)
    rel_f, rel_p = _rel_l2(g_feat.cpu().numpy(), o_gf), _rel_l2(g_pts.cpu().numpy(), o_gp)
    assert rel_f <= 1e-3 and rel_p <= 1e-3, (rel_f, rel_p)
    gp = g_pts.cpu().numpy()
    assert np.isfinite(gp).all() and np.all(gp[~o_vis] == 0)
    if P > 2_000_000:
        # renderer-owned cached point order (DSS_WS_ORDER_SAVE / _REUSE, above 2M points: configs 4 and 5): the call that
        # saves the order and two calls that reuse it give the fragments the oracle was just compared with, bit for bit
        for _ in range(3):
            fo = ops.render_forward(t(pts), t(nrm), torch.full((N,), h, device=DEV), t(M), t(V),
                                    torch.full((N,), 0.1, device=DEV), torch.full((N,), 100.0, device=DEV), first, num,
                                    t(np.tile(col, (N, 1))), S, K, CUTOFF, THR, SIGMA, False, True, order_refresh=3)
            for key in ("idx", "zbuf", "qvalue", "occupancy", "visible", "image", "wsum"):
                assert torch.equal(fo[key], f[key]), ("cached point order", key)
        del fo
    # the unfused entry points agree with the fused ones bit for bit on the fragments
    idx2 = ops.splat_points(f["pts_screen"], f["ellipse_params"], f["cutoff_threshold"], f["radii"], first, num, THR, S, K)[0]
    assert torch.equal(idx2, f["idx"])
    print("%s: %d splats, %.1f%% occupied, %d visible, oracle %.1fs, total %.1fs, RGBA err %.1e, grad rel %.1e / %.1e"
          % (name, P, 100 * float(o_occ.mean()), int(o_vis.sum()), t_oracle, time.time() - t0, err, rel_f, rel_p))
