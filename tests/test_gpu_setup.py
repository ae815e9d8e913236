"""GPU: fused per-point setup (culling + projection + EWA terms) and the drop-in Python API."""
import numpy as np
import pytest
import torch

import oracle
import scenes
from dss_amd import ops
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
from dss_amd.cloud import PointClouds3D
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(n_cams=3, S=128, backface=False):
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    az = [45.0 + 70.0 * k for k in range(n_cams)]
    M, V, cam = scenes.camera_matrices(1.2, 25.0, az, znear=0.6)  # near plane cuts the cloud
    col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
    return pts, nrm, col, M, V, az


@pytest.mark.parametrize("backface", [False, True])
def test_point_setup_matches_oracle(backface):
    pts, nrm, col, M, V, az = _scene()
    N, Pc, S = M.shape[0], pts.shape[0], 128
    h = scenes.global_h(pts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    out = ops.point_setup(t(pts), t(nrm), torch.full((N,), h, device=DEV), t(M), t(V),
                          torch.full((N,), 0.6, device=DEV), torch.full((N,), 100.0, device=DEV), first, num, S, 1.0, 1.0,
                          backface, True)
    sc = scenes.setup_scene(pts, nrm, M, V, S, h=h, znear=0.6, backface_culling=backface)
    valid = out["valid"].cpu().numpy()
    assert valid.sum() == sc["points"].shape[0] and 0 < valid.sum() < N * Pc
    assert np.array_equal(valid.reshape(N, Pc).sum(1), sc["num_pts"])
    # compacting the masked output reproduces the oracle's (compacted) arrays bit for bit
    for k_mine, k_or in (("pts_screen", "points"), ("ellipse_params", "ellipse"), ("radii", "radii"),
                         ("scaler", "scaler"), ("cutoff_threshold", "cutoff")):
        mine = out[k_mine].cpu().numpy()[valid]
        assert np.array_equal(mine, sc[k_or]), k_mine
    assert (out["pts_screen"][~out["valid"]][:, 2] == -1).all()


def test_project_backward_matches_torch_autograd():
    pts, nrm, col, M, V, az = _scene()
    N, Pc = M.shape[0], pts.shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    g = torch.randn(N * Pc, 3, device=DEV)
    valid = torch.rand(N * Pc, device=DEV) > 0.2
    gw = ops.project_backward(t(pts), t(M), t(V), first, num, g, valid, True)
    # fp64 torch reference of the same projection
    w = t(pts).double().requires_grad_(True)
    ph = torch.cat([w, torch.ones(Pc, 1, device=DEV, dtype=torch.float64)], 1)
    clip = ph[None] @ t(M).double()
    zv = (ph[None] @ t(V).double())[..., 2]
    scr = torch.stack([clip[..., 0] / clip[..., 3], clip[..., 1] / clip[..., 3], zv], -1).reshape(N * Pc, 3)
    (scr * (g * valid[:, None]).double()).sum().backward()
    rel = (gw.double() - w.grad).norm() / w.grad.norm()
    assert rel < 1e-5


@pytest.mark.parametrize("n_cams", [1, 3])
def test_renderer_api_forward_backward_vs_oracle(n_cams):
    """The drop-in classes (same constructor / forward signatures as DSS.core.rasterizer / renderer)."""
    S, K = 128, 5
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    h = scenes.global_h(pts)
    col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
    az = [45.0 + 100.0 * k for k in range(n_cams)]
    R, T = look_at_view_transform(2.0, 30.0, az)
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    settings = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                           radii_backward_scaler=5, image_size=S, points_per_pixel=K, bin_size=None,
                                           clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=settings), NormWeightedCompositor())
    P = torch.nn.Parameter(torch.from_numpy(pts).to(DEV))
    C = torch.nn.Parameter(torch.from_numpy(col).to(DEV))
    cloud = PointClouds3D([P], [torch.from_numpy(nrm).to(DEV)], [C])
    img, frags = renderer(cloud, Vrk_h=torch.tensor([h], device=DEV), verbose=True)
    assert tuple(img.shape) == (n_cams, S, S, 4) and frags.idx.dtype == torch.int32
    g = np.random.default_rng(1).standard_normal((n_cams, S, S, 4)).astype(np.float32)
    (img * torch.from_numpy(g).to(DEV)).sum().backward()

    M, V, _ = scenes.camera_matrices(2.0, 30.0, az)
    sc = scenes.setup_scene(pts, nrm, M, V, S, h=h, colors=col)
    Pn = sc["points"].shape[0]
    o_idx, o_z, o_q, o_occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"],
                                                  sc["first_idx"], sc["num_pts"], S, K, 0.05)
    assert np.array_equal(frags.idx.cpu().numpy(), o_idx)  # nothing culled here -> same packed indexing
    o_img = oracle.blend_forward(o_idx, o_q, o_occ, sc["scaler"], sc["colors"])
    assert np.abs(img.detach().cpu().numpy() - o_img).max() <= 1e-4
    o_gf, o_gocc = oracle.blend_backward(g, o_idx, o_q, sc["scaler"], Pn)
    o_gcol = o_gf.reshape(n_cams, -1, 3).sum(0)
    assert np.linalg.norm(C.grad.cpu().numpy() - o_gcol) / np.linalg.norm(o_gcol) <= 1e-3
    # position gradient: oracle screen-space gradient pushed through the fp64 projection Jacobian
    o_gs, _, _ = oracle.splat_backward(sc["points"], sc["radii"], o_idx, o_gocc, None, sc["first_idx"], sc["num_pts"],
                                       5.0, 0.05)
    w = torch.from_numpy(pts).double().requires_grad_(True)
    ph = torch.cat([w, torch.ones(w.shape[0], 1, dtype=torch.float64)], 1)
    clip = ph[None] @ torch.from_numpy(M).double()
    zv = (ph[None] @ torch.from_numpy(V).double())[..., 2]
    scr = torch.stack([clip[..., 0] / clip[..., 3], clip[..., 1] / clip[..., 3], zv], -1).reshape(-1, 3)
    (scr * torch.from_numpy(o_gs).double()).sum().backward()
    rel = np.linalg.norm(P.grad.cpu().numpy() - w.grad.numpy()) / np.linalg.norm(w.grad.numpy())
    assert rel <= 1e-3, rel


def test_foreign_compositor_call_convention():
    """pytorch3d-style call (idx (N,K,H,W) long, weights, features (C,P)) -> (N,C,H,W)."""
    sc = scenes.random_splats(500, 32, 1, seed=1)
    idx, zbuf, qv, occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                              sc["num_pts"], 32, 4, 0.5)
    w = np.where(idx >= 0, np.exp(-0.5 * qv) * sc["scaler"][np.maximum(idx, 0)], 0).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    out = NormWeightedCompositor()(t(idx).long().permute(0, 3, 1, 2), t(w).permute(0, 3, 1, 2), t(sc["colors"]).permute(1, 0))
    want = oracle.blend_forward(idx, qv, occ, sc["scaler"], sc["colors"])[..., :3]
    assert np.abs(out.permute(0, 2, 3, 1).cpu().numpy() - want).max() <= 1e-4


@pytest.mark.parametrize("shared", [True, False])
def test_render_forward_fused_equals_separate_calls(shared):
    pts, nrm, col, M, V, az = _scene()
    N, Pc, S, K = M.shape[0], pts.shape[0], 128, 5
    h = scenes.global_h(pts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    if shared:
        world, normals = t(pts), t(nrm)
    else:
        world, normals = t(np.tile(pts, (N, 1))), t(np.tile(nrm, (N, 1)))
    feat = t(np.tile(col, (N, 1)))
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    hh = torch.full((N,), h, device=DEV)
    zn, zf = torch.full((N,), 0.6, device=DEV), torch.full((N,), 100.0, device=DEV)
    for rows in (None, (32, 96)):
        f = ops.render_forward(world, normals, hh, t(M), t(V), zn, zf, first, num, feat, S, K, 1.0, 0.05, 1.0, True,
                               shared, rows=rows)
        info = ops.point_setup(world, normals, hh, t(M), t(V), zn, zf, first, num, S, 1.0, 1.0, True, shared)
        idx, zbuf, qv, occ, vis = ops.splat_points(info["pts_screen"], info["ellipse_params"], info["cutoff_threshold"],
                                                   info["radii"], first, num, 0.05, S, K, None, None, rows=rows,
                                                   return_visible=True)
        img, wsum = ops.blend_forward(idx, qv, occ, info["scaler"], feat, return_wsum=True)
        for k in ("pts_screen", "ellipse_params", "radii", "scaler", "cutoff_threshold", "valid"):
            assert torch.equal(f[k], info[k]), k
        assert torch.equal(f["idx"], idx) and torch.equal(f["zbuf"], zbuf) and torch.equal(f["qvalue"], qv)
        assert torch.equal(f["occupancy"], occ) and torch.equal(f["visible"], vis)
        assert torch.equal(f["image"], img) and torch.equal(f["wsum"], wsum)


@pytest.mark.parametrize("n_cams", [1, 3])
def test_fused_renderer_matches_unfused(n_cams):
    """SurfaceSplattingRenderer(fused=True): one autograd node on the fused kernels."""
    S, K = 128, 5
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    h = scenes.global_h(pts)
    col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
    R, T = look_at_view_transform(2.0, 30.0, [45.0 + 100.0 * k for k in range(n_cams)])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=K, bin_size=None,
                                     clip_pts_grad=0.05)
    g = torch.from_numpy(np.random.default_rng(1).standard_normal((n_cams, S, S, 4)).astype(np.float32)).to(DEV)
    res = []
    for fused in (False, True):
        renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                            fused=fused)
        P = torch.nn.Parameter(torch.from_numpy(pts).to(DEV))
        C = torch.nn.Parameter(torch.from_numpy(col).to(DEV))
        img, frags = renderer(PointClouds3D([P], [torch.from_numpy(nrm).to(DEV)], [C]),
                              Vrk_h=torch.tensor([h], device=DEV), verbose=True)
        (img * g).sum().backward()
        res.append((img.detach(), frags.idx, P.grad.clone(), C.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(res[1][2], res[0][2]) < 1e-5 and rel(res[1][3], res[0][3]) < 1e-5


@pytest.mark.parametrize("n_cams", [1, 2])
def test_graphed_renderer_matches_eager_over_parameter_updates(n_cams):
    """SurfaceSplattingRenderer(graphed=True): forward / backward replayed as two hipGraphs that read the parameters in
    place -- three iterations with in-place updates between them must give the eager fused renderer's images and gradients
    bit for bit (same kernels, same inputs), and a changed tensor address must re-capture instead of reading stale memory."""
    S, K = 128, 5
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    h = torch.tensor([scenes.global_h(pts)], device=DEV)
    col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
    R, T = look_at_view_transform(2.0, 30.0, [45.0 + 100.0 * k for k in range(n_cams)])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=K, bin_size=None,
                                     clip_pts_grad=0.05)
    g = torch.from_numpy(np.random.default_rng(1).standard_normal((n_cams, S, S, 4)).astype(np.float32)).to(DEV)
    nrm_t = torch.from_numpy(nrm).to(DEV)
    runs = {}
    for graphed in (False, True):
        renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                            fused=True, graphed=graphed)
        P = torch.nn.Parameter(torch.from_numpy(pts).to(DEV))
        C = torch.nn.Parameter(torch.from_numpy(col).to(DEV))
        out = []
        for it in range(4):
            if it == 3:   # a new tensor (new address) for the colours: the graphed mode has to notice
                C = torch.nn.Parameter(C.detach().clone() * 0.5)
            P.grad = C.grad = None
            img = renderer(PointClouds3D([P], [nrm_t], [C]), Vrk_h=h)
            (img * g).sum().backward()
            out.append((img.detach().clone(), P.grad.clone(), C.grad.clone()))
            with torch.no_grad():   # in-place update, like an optimiser step
                P -= 1e-3 * P.grad
                C -= 1e-2 * C.grad
        runs[graphed] = out
    for a, b in zip(runs[False], runs[True]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert not torch.equal(runs[True][0][0], runs[True][1][0])   # the updates did change the render


def test_fused_renderer_backward_with_64_bit_gather_falls_back_to_the_separate_projection():
    """with DSS_OPT_BACKWARD_ADDR64 (or gathered tensors of 4 GB and more) the gather kernel has no fused
    projection epilogue; the autograd node must then run the separate projection kernel, not raise inside backward."""
    from dss_amd import _lib
    S, K = 96, 5
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    h = scenes.global_h(pts)
    col = np.random.default_rng(0).uniform(0, 1, pts.shape).astype(np.float32)
    R, T = look_at_view_transform(2.0, 30.0, [45.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=K, bin_size=None,
                                     clip_pts_grad=0.05)
    g = torch.from_numpy(np.random.default_rng(1).standard_normal((1, S, S, 4)).astype(np.float32)).to(DEV)
    res = []
    for addr64 in (0, 1):
        old = _lib.set_option(_lib.OPT_BACKWARD_ADDR64, addr64)
        try:
            renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                                fused=True)
            P = torch.nn.Parameter(torch.from_numpy(pts).to(DEV))
            C = torch.nn.Parameter(torch.from_numpy(col).to(DEV))
            img = renderer(PointClouds3D([P], [torch.from_numpy(nrm).to(DEV)], [C]), Vrk_h=torch.tensor([h], device=DEV))
            (img * g).sum().backward()
            res.append((P.grad.clone(), C.grad.clone()))
        finally:
            _lib.set_option(_lib.OPT_BACKWARD_ADDR64, old)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(res[1][0], res[0][0]) < 1e-5 and rel(res[1][1], res[0][1]) < 1e-5


@pytest.mark.parametrize("mode", ["global", "iso"])
@pytest.mark.parametrize("tag", ["1cam", "3cam"])
def test_point_setup_matches_reference_python_golden(golden_dir, tag, mode):
    """HIP per-point setup vs vectors produced by the reference's own Python (make_golden_setup.py)."""
    import os
    from test_oracle_pinning import _check_setup_against_reference, _setup_inputs
    z = np.load(os.path.join(golden_dir, "ref_setup_teapot.npz"))
    pw, nw, h, cloud_of, M, V = _setup_inputs(z, tag, mode)
    N, Pc = M.shape[0], len(z["points"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    out = ops.point_setup(t(pw), t(nw), t(h), t(M), t(V), torch.full((N,), 0.1, device=DEV),
                          torch.full((N,), 100.0, device=DEV), first, num, int(z["S"]), 1.0, 1.0, False, False)
    assert out["valid"].all()
    _check_setup_against_reference(z, tag, mode, out["radii"].cpu().numpy(), out["ellipse_params"].cpu().numpy(),
                                   out["scaler"].cpu().numpy(), out["cutoff_threshold"].cpu().numpy())


@pytest.mark.parametrize("h_scale", [1.0, 40.0])
def test_render_forward_leaves_its_workspace_clean(h_scale):
    """DSS_WS_CLEAN contract (include/dss_hip.h): ops.render_forward skips the counter memset and relies on the
    fine pass to zero every tile counter, queue slot and flag it has read.  Repeated calls must give identical
    bits, and the zero region of the cached buffer must be all zero after each call.  h_scale = 40 makes giant
    splats: dense tiles (heavy-first queue in use) and overflowing sub-lists (cloud-scan fallback)."""
    from dss_amd import _lib
    pts, nrm, col, M, V, az = _scene()
    N, Pc, S, K = M.shape[0], pts.shape[0], 128, 5
    h = scenes.global_h(pts) * h_scale
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    world, normals, feat = t(pts), t(nrm), t(np.tile(col, (N, 1)))
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    hh = torch.full((N,), h, device=DEV)
    zn, zf = torch.full((N,), 0.6, device=DEV), torch.full((N,), 100.0, device=DEV)
    zero_bytes = _lib.load().dss_splat_forward_clean_bytes(N, N * Pc, S)  # the region the contract covers
    assert zero_bytes >= N * (S // 8) ** 2 * 8 * 4
    outs = []
    for rep in range(3):
        f = ops.render_forward(world, normals, hh, t(M), t(V), zn, zf, first, num, feat, S, K, 1.0, 0.05, 1.0, True, True)
        torch.cuda.synchronize()
        bufs = [b for k, b in _lib._clean_cache.items() if k[2] == ("render_forward", N, N * Pc, S)]
        assert len(bufs) == 1 and int(bufs[0][:zero_bytes].count_nonzero()) == 0, rep
        outs.append(f)
    for k in ("idx", "zbuf", "qvalue", "occupancy", "image", "wsum", "visible"):
        assert torch.equal(outs[0][k], outs[1][k]) and torch.equal(outs[0][k], outs[2][k]), k
    info = ops.point_setup(world, normals, hh, t(M), t(V), zn, zf, first, num, S, 1.0, 1.0, True, True)
    idx, zbuf, qv, occ, vis = ops.splat_points(info["pts_screen"], info["ellipse_params"], info["cutoff_threshold"],
                                               info["radii"], first, num, 0.05, S, K, None, None, return_visible=True)
    assert torch.equal(outs[2]["idx"], idx) and torch.equal(outs[2]["zbuf"], zbuf) and torch.equal(outs[2]["visible"], vis)


def test_renderer_without_compositor_is_unnormalised_weighted_sum():
    """renderer.py:59-65: ``compositor=None`` -> pytorch3d ``weighted_sum`` (sum_k f_k w_k, no normalisation, no
    gradient to the weights).  Checked against the same sum evaluated with plain torch ops on the fragments."""
    S, K = 96, 4
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    h = scenes.global_h(pts)
    col = np.random.default_rng(3).uniform(0, 1, pts.shape).astype(np.float32)
    R, T = look_at_view_transform(2.0, 30.0, [45.0, 160.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    settings = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, Vrk_invariant=True,
                                           radii_backward_scaler=5, image_size=S, points_per_pixel=K, bin_size=None,
                                           clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=settings), None)
    C = torch.nn.Parameter(torch.from_numpy(col).to(DEV))
    cloud = PointClouds3D([torch.from_numpy(pts).to(DEV)], [torch.from_numpy(nrm).to(DEV)], [C])
    img, fr = renderer(cloud, Vrk_h=torch.tensor([h], device=DEV), verbose=True)
    feat = torch.cat([C, C], 0)  # camera-extended packed features, the order fragments.idx refers to
    feat2 = feat.detach().clone().requires_grad_(True)
    valid = fr.idx >= 0
    safe = fr.idx.clamp_min(0).long()
    w = torch.exp(-0.5 * fr.qvalue) * fr.scaler_packed[safe] * valid
    # the per-fragment view has the reference's shape and values (rasterizer.py:631-633: 0 where idx < 0)
    assert fr.scaler.shape == fr.qvalue.shape and torch.equal(fr.scaler, fr.scaler_packed[safe] * valid)
    assert len(fr) == 5 and [tuple(t.shape) for t in fr][3] == tuple(fr.qvalue.shape) and fr._fields[3] == "scaler"
    want = (feat2[safe] * w.unsqueeze(-1)).sum(dim=3)
    assert torch.allclose(img[..., :3], want, rtol=1e-4, atol=1e-5)
    assert torch.equal(img[..., 3], fr.occupancy)
    g = torch.randn_like(img)
    (img * g).sum().backward()
    (want * g[..., :3]).sum().backward()
    g_ref = feat2.grad[:pts.shape[0]] + feat2.grad[pts.shape[0]:]
    assert (C.grad - g_ref).norm() / g_ref.norm() <= 1e-4


def test_local_frames_match_oracle_and_reference_golden(golden_dir):
    """dss_local_frames (fp32 Jacobi) vs the oracle's fp64 restatement and vs the reference's own
    `_compute_anisotropic_Vrk` (golden, make_golden_setup.py); neighbourhoods from dss_knn_points (K = 8)."""
    import os
    z = np.load(os.path.join(golden_dir, "ref_setup_teapot.npz"))
    pts = z["points"]
    P = len(pts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.zeros(1, dtype=torch.int64, device=DEV)
    num = torch.full((1,), P, dtype=torch.int64, device=DEV)
    _, idx = ops.knn_points(t(pts), first, num, 8)
    vr6, fn, cv = ops.local_frames(t(pts), idx, first, num, return_curvature=True)
    o_vr6, o_fn, o_cv = oracle.local_frames(pts, idx.cpu().numpy())
    scale = np.abs(o_vr6).max()
    gap = (o_cv[:, 1] - o_cv[:, 0]) / np.maximum(o_cv[:, 2], 1e-30) > 1e-2   # well-defined normal
    assert np.abs(vr6.cpu().numpy() - o_vr6)[gap].max() <= 1e-4 * scale
    assert np.allclose(cv.cpu().numpy(), o_cv, rtol=1e-3, atol=1e-5 * o_cv.max())
    assert (np.abs((fn.cpu().numpy() * o_fn).sum(1))[gap] >= 1 - 1e-4).all()     # same normal up to sign
    ref_vr = z["aniso_Vr"]
    got = vr6.cpu().numpy()
    got33 = np.stack([got[:, [0, 1, 2]], got[:, [1, 3, 4]], got[:, [2, 4, 5]]], 1)
    assert np.abs(got33 - ref_vr)[gap].max() <= 3e-4 * np.abs(ref_vr).max()


@pytest.mark.parametrize("tag", ["1cam", "3cam"])
def test_anisotropic_rasterizer_matches_reference_python_golden(golden_dir, tag):
    """End to end through the drop-in class: SurfaceSplatting with Vrk_invariant = Vrk_isotropic = False (kNN-8 ->
    PCA frames -> fused setup) against the reference's `_get_per_point_info` in the same mode."""
    import os
    z = np.load(os.path.join(golden_dir, "ref_setup_teapot.npz"))
    pts, nrm = z["points"], z["normals"]
    az = {"1cam": [45.0], "3cam": [10.0, 130.0, 250.0]}[tag]
    R, T = look_at_view_transform(2.0, 30.0, az)
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    N, S = len(az), int(z["S"])
    st = PointsRasterizationSettings(cutoff_threshold=1.0, image_size=S, antialiasing_sigma=1.0, Vrk_invariant=False,
                                     Vrk_isotropic=False, points_per_pixel=5, backface_culling=False)
    rast = SurfaceSplatting(cameras=cams, raster_settings=st)
    cloud = PointClouds3D([torch.from_numpy(pts).to(DEV)], [torch.from_numpy(nrm).to(DEV)],
                          [torch.ones(len(pts), 3, device=DEV)])
    frags, out_cloud, info = rast(cloud, verbose=True)
    ref = lambda k: z["%s_aniso_%s" % (tag, k)]
    o_vr6, o_fn, o_cv = oracle.local_frames(pts, np.asarray(
        __import__("scipy.spatial", fromlist=["cKDTree"]).cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=8)[1]))
    gap = np.tile((o_cv[:, 1] - o_cv[:, 0]) / np.maximum(o_cv[:, 2], 1e-30), N) > 1e-2
    ra, el, sc = (info[k].cpu().numpy() for k in ("radii", "ellipse_params", "scaler"))
    assert np.allclose(ra[gap], ref("radii")[gap], rtol=5e-4, atol=0)
    assert np.abs(el - ref("ellipse_params"))[gap].max() <= 5e-4 * np.abs(ref("ellipse_params")).max()
    assert np.abs(sc - ref("scaler"))[gap].max() <= 5e-4 * np.abs(ref("scaler")).max()
    assert frags.occupancy.mean().item() > 0.05
    # the fused renderer takes the same branch
    img_a = SurfaceSplattingRenderer(rast, NormWeightedCompositor(), fused=True)(cloud)
    img_b = SurfaceSplattingRenderer(rast, NormWeightedCompositor(), fused=False)(cloud)
    assert torch.equal(img_a, img_b)


def test_anisotropic_point_setup_matches_oracle_bits():
    """Same Vrk in -> HIP setup == oracle setup bit for bit (like the other two variance modes)."""
    pts, nrm, col, M, V, az = _scene()
    N, Pc, S = M.shape[0], pts.shape[0], 128
    from scipy.spatial import cKDTree
    _, nn = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=8)
    vr6, fn, _ = oracle.local_frames(pts, nn)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    out = ops.point_setup(t(pts), t(nrm), torch.zeros(Pc, device=DEV), t(M), t(V), torch.full((N,), 0.01, device=DEV),
                          torch.full((N,), 100.0, device=DEV), first, num, S, 1.0, 1.0, False, True, vr6=t(vr6),
                          frame_normals=t(fn))
    assert out["valid"].all()
    tile = lambda a: np.tile(a, (N, 1))
    ps, el, ra, sc, cu = oracle.point_setup(tile(pts), tile(nrm), np.zeros(N * Pc, np.float32),
                                            np.repeat(np.arange(N, dtype=np.int32), Pc), M, V, S, 1.0, 1.0,
                                            vr6=tile(vr6), frame_normals=tile(fn))
    for mine, want in ((out["pts_screen"], ps), (out["ellipse_params"], el), (out["radii"], ra), (out["scaler"], sc)):
        assert np.array_equal(mine.cpu().numpy(), want)


def test_project_backward_fused_clip_equals_clip_then_project():
    pts, nrm, col, M, V, az = _scene()
    N, Pc = M.shape[0], pts.shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    g = torch.randn(N * Pc, 3, device=DEV) * 0.1
    valid = torch.rand(N * Pc, device=DEV) > 0.2
    fused = ops.project_backward(t(pts), t(M), t(V), first, num, g, valid, True, clip=0.05)
    g2 = g.clone()
    ops.clip_grad_(g2, 0.05)
    assert not torch.equal(g2, g)
    want = ops.project_backward(t(pts), t(M), t(V), first, num, g2, valid, True)
    assert torch.equal(fused, want)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 3, 8, 11])
def test_project_backward_reduces_the_feature_gradients_of_a_shared_cloud(N):
    """`dss_project_backward_features`: the per-camera feature gradients of a cloud shared by N cameras summed over the
    cameras in the launch that projects the position gradients (more than eight cameras take a second batch of loads);
    the position gradients are those of dss_project_backward bit for bit, with and without the clip."""
    pts = scenes.normalize_unit_sphere(scenes.load_cloud("bunny")[0])[:3001]
    Pc = pts.shape[0]
    mats = [scenes.camera_matrices(2.0, 20.0, 360.0 * k / N + 10.0) for k in range(N)]
    M, V = np.concatenate([m[0] for m in mats]), np.concatenate([m[1] for m in mats])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV) * Pc
    num = torch.full((N,), Pc, device=DEV, dtype=torch.int64)
    g = torch.randn(N * Pc, 3, device=DEV) * 0.1
    valid = torch.rand(N * Pc, device=DEV) > 0.2
    for C in (3, 5):
        gf = torch.randn(N * Pc, C, device=DEV)
        for clip in (-1.0, 0.05):
            want = ops.project_backward(t(pts), t(M), t(V), first, num, g, valid, True, clip=clip)
            gw, gfw = ops.project_backward(t(pts), t(M), t(V), first, num, g, valid, True, clip=clip, grad_features=gf)
            assert torch.equal(gw, want)
            ref = gf.view(N, Pc, C).double().sum(0)
            assert float((gfw.double() - ref).abs().max()) <= 1e-5 * max(float(ref.abs().max()), 1.0)
    # clouds that are not shared: the "reduction" is the identity
    gw, gfw = ops.project_backward(t(np.tile(pts, (N, 1))), t(M), t(V), first, num, g, valid, False, grad_features=gf)
    assert torch.equal(gfw, gf) and torch.equal(gw, ops.project_backward(t(np.tile(pts, (N, 1))), t(M), t(V), first, num, g, valid, False))


def test_filters_object_drops_inactive_points_and_receives_visibility():
    """`point_clouds_filter` (DSS/core/cloud.py:284-351) through the renderer: inactive points are not rendered
    (rasterizer.py:230-234) and the per-point visibility comes back as a padded (N, P_max) mask over the ORIGINAL
    cloud (rasterizer.py:639-652), equal to the visibility of rendering the reduced cloud scattered back."""
    from dss_amd.cloud import PointCloudsFilters
    from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
    from dss_amd.cloud import PointClouds3D
    from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    P = pts.shape[0]
    rng = np.random.default_rng(11)
    act = torch.from_numpy(rng.random(P) < 0.7).to(DEV)
    R, T = look_at_view_transform(2.0, 25.0, [20.0, 140.0, 260.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, image_size=96, points_per_pixel=5, bin_size=None)
    tp, tn = torch.from_numpy(pts).to(DEV), torch.from_numpy(nrm).to(DEV)
    col = torch.rand(P, 3, device=DEV)
    for fused in (False, True):
        renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                            fused=fused)
        flt = PointCloudsFilters(device=DEV, activation=act[None])
        img = renderer(PointClouds3D([tp], [tn], [col]), point_clouds_filter=flt)
        ref_flt = PointCloudsFilters(device=DEV)
        ref = renderer(PointClouds3D([tp[act]], [tn[act]], [col[act]]), point_clouds_filter=ref_flt)
        assert torch.equal(img, ref)
        assert tuple(flt.visibility.shape) == (3, P) and flt.visibility.dtype == torch.bool
        assert not flt.visibility[:, ~act].any() and flt.visibility.any()
        assert tuple(ref_flt.visibility.shape) == (3, int(act.sum()))
        assert torch.equal(flt.visibility[:, act], ref_flt.visibility)


@pytest.mark.parametrize("fused", [False, True])
def test_replicated_clouds_take_the_shared_geometry_path_with_identical_results(fused):
    """`Pointclouds.extend(N)` = the same position / normal tensors N times with per-camera colours (what the texture
    hands to the renderer): the geometry is processed once (shared-cloud kernels, one kNN); images and gradients must
    equal those of N physically separate copies."""
    from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
    from dss_amd.cloud import PointClouds3D
    from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
    from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    R, T = look_at_view_transform(2.0, 25.0, [20.0, 140.0, 260.0])
    cams = FoVPerspectiveCameras(znear=0.1, R=R, T=T, device=DEV)
    st = PointsRasterizationSettings(backface_culling=False, image_size=96, points_per_pixel=5, bin_size=None,
                                     radii_backward_scaler=5, clip_pts_grad=0.05)
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(),
                                        fused=fused)
    cols = [torch.rand(pts.shape[0], 3, device=DEV) for _ in range(3)]
    g = torch.randn(3, 96, 96, 4, device=DEV)
    p1 = torch.from_numpy(pts).to(DEV).requires_grad_(True)
    n1 = torch.from_numpy(nrm).to(DEV)
    img1 = renderer(PointClouds3D([p1, p1, p1], [n1, n1, n1], cols))
    (img1 * g).sum().backward()
    p2 = torch.from_numpy(pts).to(DEV).requires_grad_(True)
    img2 = renderer(PointClouds3D([p2 * 1.0, p2 * 1.0, p2 * 1.0], [n1.clone(), n1.clone(), n1.clone()], cols))
    (img2 * g).sum().backward()
    assert torch.equal(img1, img2)
    assert (p1.grad - p2.grad).norm() <= 1e-5 * p2.grad.norm()


def test_compact_culled_mode_reproduces_the_reference_labels_under_culling():
    """Depth-range + back-face culling: by default culled points are masked (labels index the un-filtered cloud); with
    ``compact_culled=True`` the rasterizer drops them first like rasterizer.py:219-254, computes h on the filtered cloud
    (:310-326) and returns that cloud -- `fragments.idx` then equals the oracle's (which compacts the same way) integer
    for integer, and both modes give the same image."""
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    S, K = 96, 5
    M, V, _ = scenes.camera_matrices(2.0, 20.0, 30.0)
    znear = 1.9                                   # cuts the front of the object away
    keep = ((np.concatenate([pts, np.ones((len(pts), 1), np.float32)], 1) @ V[0])[:, 2] >= znear) & \
        ((nrm @ V[0][:3, :3])[:, 2] < 0)
    assert 0.2 < keep.mean() < 0.6
    h = scenes.global_h(pts[keep])               # the reference runs its kNN on the FILTERED cloud
    sc = scenes.setup_scene(pts, nrm, M, V, S, h=h, znear=znear, backface_culling=True)
    want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"], sc["num_pts"],
                                S, K, 0.05)
    from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
    R, T = look_at_view_transform(2.0, 20.0, 30.0)
    cams = FoVPerspectiveCameras(R=R, T=T, znear=znear, zfar=100.0, device=DEV)
    st = PointsRasterizationSettings(backface_culling=True, cutoff_threshold=1.0, Vrk_invariant=True, Vrk_isotropic=False,
                                     radii_backward_scaler=5, image_size=S, points_per_pixel=K, bin_size=None)
    col = torch.rand(len(pts), 3, device=DEV)
    cloud = PointClouds3D([torch.from_numpy(pts).to(DEV)], [torch.from_numpy(nrm).to(DEV)], [col])
    strict = SurfaceSplatting(cameras=cams, raster_settings=st, compact_culled=True)
    fr, out_cloud, info = strict(cloud, verbose=True)
    assert out_cloud.points_packed().shape[0] == int(keep.sum())
    assert np.array_equal(out_cloud.points_packed().cpu().numpy(), pts[keep])
    assert abs(float(strict._Vrk_h.flatten()[0]) - h) <= 1e-6 * h
    assert np.array_equal(fr.idx.cpu().numpy(), want[0]) and np.array_equal(fr.occupancy.cpu().numpy(), want[3])
    assert info["radii"].shape[0] == len(pts) and float(info["radii"][~torch.from_numpy(keep).to(DEV)].abs().max()) == 0
    # default (compact_culled=None) follows raster_settings.backface_culling: a default-constructed rasterizer gives the
    # reference's labels and image without being told
    default = SurfaceSplatting(cameras=cams, raster_settings=st)
    assert default.compacts() and not SurfaceSplatting(cameras=cams, raster_settings=PointsRasterizationSettings(
        backface_culling=False)).compacts() and SurfaceSplatting(cameras=cams).compacts()
    fr_d, cloud_d = default(cloud)
    assert torch.equal(fr_d.idx, fr.idx) and cloud_d.points_packed().shape[0] == int(keep.sum())
    masked = SurfaceSplatting(cameras=cams, raster_settings=st, compact_culled=False)
    fr_m, cloud_m = masked(cloud, Vrk_h=torch.tensor([h], device=DEV))
    remap = torch.from_numpy(np.nonzero(keep)[0]).to(DEV)
    assert torch.equal(torch.where(fr.idx >= 0, remap[fr.idx.clamp_min(0).long()].int(), fr.idx), fr_m.idx)
    img_s = SurfaceSplattingRenderer(strict, None)(cloud)
    img_m = SurfaceSplattingRenderer(masked, None)(cloud, Vrk_h=torch.tensor([h], device=DEV))
    assert float((img_s - img_m).abs().max()) <= 1e-5 * max(1.0, float(img_m.abs().max()))
