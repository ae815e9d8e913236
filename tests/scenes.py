"""Seeded scene builders shared by the tests, __graft_entry__.smoke() and bench.py's CPU leg.

TEST INFRASTRUCTURE: uses the oracle for the per-point EWA setup so that the inputs handed to the
HIP rasterizer are independent of the product's own setup code.
"""
import os

import numpy as np

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cloud(name: str):
    """-> (points (P,3), normals (P,3)) float32 from tests/golden/clouds.npz
    (converted from the reference's example_data/pointclouds/*.ply by make_golden.py)."""
    z = np.load(os.path.join(_GOLDEN, "clouds.npz"))
    return z[name + "_points"].astype(np.float32), z[name + "_normals"].astype(np.float32)


def normalize_unit_sphere(points):
    c = (points.max(0) + points.min(0)) / 2
    p = points - c
    return (p / np.linalg.norm(p, axis=1).max()).astype(np.float32)


def upsample_jitter(points, normals, factor: int, seed: int = 0):
    """Deterministic xfactor upsample: every point spawns `factor` copies displaced in its tangent
    plane by N(0, 0.5*mean nearest-neighbour spacing) (SURVEY 8d, cfg2 'bunny ~30k')."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    d, _ = cKDTree(points).query(points, k=2)
    spacing = float(d[:, 1].mean())
    n = normals / np.maximum(np.linalg.norm(normals, axis=1, keepdims=True), 1e-12)
    helper = np.where(np.abs(n[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    u = np.cross(n, helper)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(n, u)
    outs = [points]
    for _ in range(factor - 1):
        a = rng.normal(0, 0.5 * spacing, (points.shape[0], 2))
        outs.append(points + a[:, :1] * u + a[:, 1:] * v)
    return (np.concatenate(outs, 0).astype(np.float32), np.concatenate([normals] * factor, 0).astype(np.float32))


def synthetic_cloud(P: int, seed: int = 0):
    """Unit sphere displaced radially by 0.1 sin(4 theta) sin(4 phi), analytic-ish normals
    (SURVEY 8d cfg4/cfg5 generator) -> points, normals, colours."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(P, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    theta = np.arccos(np.clip(v[:, 2], -1, 1))
    phi = np.arctan2(v[:, 1], v[:, 0])
    rad = 1.0 + 0.1 * np.sin(4 * theta) * np.sin(4 * phi)
    pts = v * rad[:, None] * 0.9
    # normal of r(theta,phi): n ~ r*e_r - dr/dtheta e_theta - dr/dphi/sin(theta) e_phi
    dr_dt = 0.4 * np.cos(4 * theta) * np.sin(4 * phi)
    dr_dp = 0.4 * np.sin(4 * theta) * np.cos(4 * phi)
    e_t = np.stack([np.cos(theta) * np.cos(phi), np.cos(theta) * np.sin(phi), -np.sin(theta)], 1)
    e_p = np.stack([-np.sin(phi), np.cos(phi), np.zeros_like(phi)], 1)
    st = np.maximum(np.sin(theta), 1e-3)
    n = rad[:, None] * v - dr_dt[:, None] * e_t - (dr_dp / st)[:, None] * e_p
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    col = rng.uniform(0, 1, (P, 3))
    return pts.astype(np.float32), n.astype(np.float32), col.astype(np.float32)


def global_h(points) -> float:
    """Vrk_invariant variance scale: mean over the cloud of 0.5*max(kNN-7 squared distances),
    clamped to [5e-5, 1e-3] (rasterizer.py:310-326)."""
    from scipy.spatial import cKDTree
    if points.shape[0] < 7:
        return float(np.clip(0.5e-3, 5e-5, 1e-3))
    d, _ = cKDTree(points).query(points, k=7)
    sq = (d[:, 1:] ** 2).astype(np.float32)
    return float(np.clip((0.5 * sq.max(1)).mean(), 5e-5, 1e-3))


def large_cloud_h(points) -> float:
    """Variance scale h of the synthetic large clouds (BASELINE configs[3] / configs[4]): the kNN-7 statistic of a 200k-point
    subsample scaled by the density ratio (the reference's clamp [5e-5, 1e-3] would turn a 4M-point cloud into 20-pixel
    splats), clipped to [5e-6, 1e-3].  ONE definition for `bench.py::large_cloud` and `tests/test_gpu_named_configs.py`: the
    benched configuration is the parity-tested one."""
    P = points.shape[0]
    h = global_h(points[:: max(1, P // 200_000)]) * (200_000 / P if P > 200_000 else 1.0)
    return float(np.clip(h, 5e-6, 1e-3))


# cameras of the bench workloads: ring at distance 2, elevation 30 degrees, azimuth 45 + 45 k (bench.py Workload)
BENCH_CAMERA = (2.0, 30.0)


def bench_azimuths(n):
    return [45.0 + 45.0 * k for k in range(n)]


def camera_matrices(dist, elev, azim, znear=0.1, zfar=100.0, fov=60.0):
    """-> M (N,4,4) full projection, V (N,4,4) world->view (row-vector convention), float32."""
    from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
    R, T = look_at_view_transform(dist, elev, azim)
    cam = FoVPerspectiveCameras(znear=znear, zfar=zfar, fov=fov, R=R, T=T)
    M = cam.get_full_projection_transform().get_matrix().numpy().astype(np.float32)
    V = cam.get_world_to_view_transform().get_matrix().numpy().astype(np.float32)
    return M, V, cam


def setup_scene(points, normals, M, V, S, cutoff=1.0, sigma=1.0, h=None, colors=None,
                znear=0.1, zfar=100.0, backface_culling=False):
    """Replicates the cloud per camera, culls by view depth (+ optional back faces), runs the
    oracle's per-point EWA setup and returns packed float32 arrays (a dict)."""
    import oracle
    N = M.shape[0]
    if h is None:
        h = global_h(points)
    if colors is None:
        colors = np.ones((points.shape[0], 3), np.float32)
    pw, nw, cw, cloud_of, first, num = [], [], [], [], [], []
    off = 0
    for n in range(N):
        ph = np.concatenate([points, np.ones((points.shape[0], 1), np.float32)], 1)
        zview = (ph @ V[n])[:, 2]
        keep = (zview >= znear) & (zview <= zfar)
        if backface_culling:
            nview = normals @ V[n][:3, :3]
            keep &= nview[:, 2] < 0
        pw.append(points[keep]); nw.append(normals[keep]); cw.append(colors[keep])
        cloud_of.append(np.full(int(keep.sum()), n, np.int32))
        first.append(off); num.append(int(keep.sum())); off += int(keep.sum())
    pw, nw, cw, cloud_of = (np.concatenate(x, 0) for x in (pw, nw, cw, cloud_of))
    hh = np.full(pw.shape[0], h, np.float32)
    ps, el, ra, sc, cu = oracle.point_setup(pw, nw, hh, cloud_of, M, V, S, cutoff, sigma)
    return dict(points=ps, ellipse=el, radii=ra, scaler=sc, cutoff=cu, colors=cw.astype(np.float32),
                first_idx=np.asarray(first, np.int64), num_pts=np.asarray(num, np.int64),
                world=pw, normals=nw, cloud_of=cloud_of, h=h, S=S)


def random_splats(P, S, N=1, seed=0, negz=5, rmin=1.5, rmax=6.0, ties=False):
    """Random anisotropic splats directly in screen space (no camera): stresses the hit test."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1.05, 1.05, (N * P, 3)).astype(np.float32)
    pts[:, 2] = rng.uniform(0.5, 3, N * P).astype(np.float32)
    if ties:
        pts[:, 2] = np.round(pts[:, 2] * 4) / 4  # many exact depth ties -> exercises (z, idx) order
    if negz:
        pts[rng.integers(0, N * P, negz), 2] = -0.3
    r = rng.uniform(rmin, rmax, (N * P, 2)) * 2 / S
    th = rng.uniform(0, np.pi, N * P)
    c, s = np.cos(th), np.sin(th)
    l1, l2 = 1 / r[:, 0] ** 2, 1 / r[:, 1] ** 2
    a = c * c * l1 + s * s * l2
    cc = s * s * l1 + c * c * l2
    b = 2 * c * s * (l1 - l2)
    C = 1.0
    den = 4 * a * cc - b * b
    return dict(points=pts, ellipse=np.stack([a, b, cc], 1).astype(np.float32),
                radii=np.stack([np.sqrt(4 * cc * C / den), np.sqrt(4 * a * C / den)], 1).astype(np.float32),
                cutoff=np.full(N * P, C, np.float32),
                scaler=rng.uniform(0.5, 2.0, N * P).astype(np.float32),
                colors=rng.uniform(0, 1, (N * P, 3)).astype(np.float32),
                first_idx=(np.arange(N) * P).astype(np.int64), num_pts=np.full(N, P, np.int64), S=S)


def self_knn(points_list, K):
    """CPU stand-in of the self query knn_points(p, p, K) on a list of clouds -> packed (dists (P,K) squared fp32,
    idx (P,K) int64 cloud-local, first_of (P,) packed id of the first point of each point's cloud); the point itself
    is entry 0, neighbours ascending."""
    from scipy.spatial import cKDTree
    d_all, i_all, f_all, first = [], [], [], 0
    for p in points_list:
        p64 = np.asarray(p, np.float64)
        _, idx = cKDTree(p64).query(p64, k=K)
        idx[:, 0] = np.arange(len(p64))  # coincident points: keep the query itself first
        d2 = ((p64[:, None, :] - p64[idx]) ** 2).sum(-1)
        d_all.append(d2.astype(np.float32)); i_all.append(idx.astype(np.int64))
        f_all.append(np.full(len(p64), first, np.int64))
        first += len(p64)
    return np.concatenate(d_all), np.concatenate(i_all), np.concatenate(f_all)
