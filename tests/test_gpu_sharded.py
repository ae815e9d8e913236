"""GPU: the row-partitioned (multi-GPU) render behind the drop-in classes -- `SurfaceSplattingRenderer(row_partition=...)`,
`dss_amd.sharded.RowShardedRender` -- and the two entry points it adds to the C ABI.

The reference has no distributed layer; what is pinned here is that the sharded render IS the single-GPU
render: the gathered image bit for bit, the gradients a training loop sees (``points.grad`` / ``colors.grad`` after a loss
computed FROM the rendered image) to fp32 reduction order.  One GPU on the test box, so two ranks share it over gloo; RCCL
itself is exercised at world size 1 by tests/test_gpu_rccl_world1.py."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_gather_rows_puts_gathered_rows_in_image_order():
    from dss_amd import ops
    from dss_amd.distributed import RowPartition
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    for S, G, N, W, cyclic in ((64, 4, 3, 64 * 4, True), (40, 4, 2, 40, False), (70, 4, 1, 70 * 3, True), (48, 2, 2, 48 * 4, False)):
        part = RowPartition(S, G, 0, cyclic=cyclic)   # (S = 70 at 4 ranks: unequal tile rows, padded positions)
        pos = part.gather_index()
        src = torch.randn((G * part.band, N, W), generator=g).to(dev)
        row_pos = torch.tensor(pos, dtype=torch.int32, device=dev)
        got = ops.gather_rows(src, row_pos, N, S, W)
        want = src.index_select(0, row_pos.long()).permute(1, 0, 2).contiguous()
        assert torch.equal(got, want), (S, G, N, W, cyclic)


@pytest.mark.parametrize("cyclic", [False, True])
def test_owner_mode_from_the_dense_alpha_plane_equals_owner_mode_from_the_full_gradient(cyclic):
    """dss_render_backward_owned_plane (the alpha channel of all rows as a dense (N,S,S) plane: what the ranks all-gather
    behind a band-local loss) against dss_render_backward_owned (the full RGBA gradient): same bits."""
    import scenes
    from dss_amd import ops
    from dss_amd.distributed import RowPartition
    dev = torch.device("cuda:0")
    S, K, N = 128, 5, 2
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    world, normals = t(pts), t(nrm)
    Pc = world.shape[0]
    col = torch.rand((N * Pc, 3), generator=torch.Generator().manual_seed(2)).to(dev)
    from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
    R, T = look_at_view_transform(2.0, 30.0, [45.0, 130.0])
    cam = FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R, T=T)
    M = cam.get_full_projection_transform().get_matrix().to(dev).contiguous()
    V = cam.get_world_to_view_transform().get_matrix().to(dev).contiguous()
    zn, zf = torch.full((N,), 0.1, device=dev), torch.full((N,), 100.0, device=dev)
    first = torch.arange(N, device=dev, dtype=torch.int64) * Pc
    num = torch.full((N,), Pc, device=dev, dtype=torch.int64)
    h = torch.full((N,), 4e-4, device=dev)
    gfull = torch.randn((N, S, S, 4), generator=torch.Generator().manual_seed(1)).to(dev)
    whole = ops.render_forward(world, normals, h, M, V, zn, zf, first, num, col, S, K, 1.0, 0.05, 1.0, False, True)
    vis_all = whole["visible"]
    for rank in range(2):
        part = RowPartition(S, 2, rank, cyclic=cyclic)
        f = ops.render_forward(world, normals, h, M, V, zn, zf, first, num, col, S, K, 1.0, 0.05, 1.0, False, True, rows=part.rows)
        gband = part.slice(gfull).contiguous()
        a = ops.render_backward(gband, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], vis_all, first,
                                num, 5.0, -1.0, image_size=S, rows=part.rows, grad_out_full=gfull)
        b = ops.render_backward(gband, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], vis_all, first,
                                num, 5.0, -1.0, image_size=S, rows=part.rows, grad_occ_full=gfull[..., 3].contiguous())
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (cyclic, rank)
        assert float(a[1].abs().max()) > 0


def test_band_outputs_only_hands_out_zeros_not_uninitialised_memory():
    """with DSS_WS_BAND_OUTPUTS the library writes ellipse / scaler / cutoff only for the splats that meet the band;
    the rest of those three arrays must be DEFINED (zero), whatever the allocator's memory held before."""
    import scenes
    from dss_amd import ops
    dev = torch.device("cuda:0")
    S, K, N = 128, 5, 1
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    world, normals = t(pts), t(nrm)
    Pc = world.shape[0]
    col = torch.rand((Pc, 3), generator=torch.Generator().manual_seed(2)).to(dev)
    Mn, Vn, _ = scenes.camera_matrices(2.0, 30.0, 45.0)
    M, V = t(Mn), t(Vn)
    zn, zf = torch.full((N,), 0.1, device=dev), torch.full((N,), 100.0, device=dev)
    first = torch.zeros(N, device=dev, dtype=torch.int64)
    num = torch.full((N,), Pc, device=dev, dtype=torch.int64)
    h = torch.full((N,), 4e-4, device=dev)
    args = (world, normals, h, M, V, zn, zf, first, num, col, S, K, 1.0, 0.05, 1.0, False, True)
    full = ops.render_forward(*args)
    for _ in range(3):   # poison the allocator's free blocks, then render a band
        junk = torch.full((Pc * 8,), float("nan"), device=dev)
        del junk
        f = ops.render_forward(*args, rows=(32, 48), band_outputs_only=True)
        for k in ("ellipse_params", "scaler", "cutoff_threshold"):
            a, b = f[k].reshape(Pc, -1), full[k].reshape(Pc, -1)
            assert bool(torch.isfinite(a).all()), k
            written = (a == b).all(dim=1)
            assert bool(((a == 0).all(dim=1) | written).all()), k       # either the splat's values or zeros
        assert bool((f["scaler"] != 0).any()) and bool((f["scaler"] == 0).any())
        assert torch.equal(f["idx"], full["idx"][:, 32:48]) and torch.equal(f["image"], full["image"][:, 32:48])


_TWO_RANK = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import scenes
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
from dss_amd.cloud import PointClouds3D
from dss_amd.distributed import RowPartition, band_image_loss
from dss_amd.losses import calc_dr_loss
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("gloo")
S, N = 256, 2
pts, nrm = scenes.load_cloud("bunny")
pts = scenes.normalize_unit_sphere(pts)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
R, T = look_at_view_transform(2.0, 30.0, [45.0, 130.0])
cams = FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R, T=T, device=dev)
st = PointsRasterizationSettings(backface_culling=False, cutoff_threshold=1.0, depth_merging_threshold=0.05, Vrk_invariant=True,
                                 Vrk_isotropic=False, radii_backward_scaler=5.0, image_size=S, points_per_pixel=5,
                                 bin_size=None, clip_pts_grad=0.05, antialiasing_sigma=1.0)
normals = t(nrm)
col0 = torch.rand((pts.shape[0], 3), generator=torch.Generator().manual_seed(3)).to(dev)
h = torch.full((1,), 4e-4, device=dev)


def run(**kw):
    """one training iteration through the classes: render -> loss FROM the rendered image -> backward"""
    X = torch.nn.Parameter(t(pts).clone())
    C = torch.nn.Parameter(col0.clone())
    renderer = SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams, raster_settings=st), NormWeightedCompositor(), **kw)
    img = renderer(PointClouds3D([X], [normals], [C]), Vrk_h=h)
    return renderer, X, C, img


# single rank: the plain renderer; targets = its own render shifted (identical on both ranks)
_, X1, C1, img1 = run()
target = torch.roll(img1.detach(), shifts=(4, 7), dims=(1, 2))
target_rgb, target_mask = target[..., :3].contiguous(), (target[..., 3] > 0).float().contiguous()
loss1 = calc_dr_loss(img1, target_rgb, target_mask)["loss"]
loss1.backward()
rel = lambda a, b: float((a - b).norm() / b.norm())
assert float(X1.grad.abs().max()) > 0 and float(C1.grad.abs().max()) > 0
report = []
for cyclic in (False, True):
    part = RowPartition(S, world, rank, cyclic=cyclic)
    for grad in ("owner", "bucket"):
        # replicated loss: the FULL image on every rank, the loss evaluated on it like an unmodified training loop
        _, X, C, img = run(row_partition=part, gradient_exchange=grad)
        assert img.shape == img1.shape and torch.equal(img, img1.detach()), ("full image differs", cyclic, grad)
        loss = calc_dr_loss(img, target_rgb, target_mask)["loss"]
        assert abs(float(loss) - float(loss1)) <= 1e-6 * abs(float(loss1))
        loss.backward()
        r = (rel(X.grad, X1.grad), rel(C.grad, C1.grad))
        assert r[0] < 1e-5 and r[1] < 1e-5, ("full", cyclic, grad, r)
        report.append(("full", cyclic, grad) + r)
        # band loss: the rank's own rows, the reference's image loss with its sums all-reduced
        ren, X, C, band = run(row_partition=part, gradient_exchange=grad, row_output="band")
        assert tuple(band.shape) == (N, part.n_rows, S, 4) and torch.equal(band, part.slice(img1.detach()))
        loss = band_image_loss(band, target_rgb, target_mask, part)["loss"]
        assert abs(float(loss) - float(loss1)) <= 1e-6 * abs(float(loss1)), (float(loss), float(loss1))
        loss.backward()
        r = (rel(X.grad, X1.grad), rel(C.grad, C1.grad))
        assert r[0] < 1e-5 and r[1] < 1e-5, ("band", cyclic, grad, r)
        report.append(("band", cyclic, grad) + r)
        # the image exchange of the band render ran behind the loss and the backward: the full picture is there for whoever looks
        eng = next(iter(ren.rasterizer._sharded.values()))
        assert torch.equal(eng.full_image(), img1.detach())
# "auto": this rank's share of the initialised process group; two renders in flight (the second must not disturb the first)
_, X, C, img = run(row_partition="auto")
_, X2, C2, img2 = run(row_partition="auto")
calc_dr_loss(img, target_rgb, target_mask)["loss"].backward()
assert rel(X.grad, X1.grad) < 1e-5 and rel(C.grad, C1.grad) < 1e-5
open(os.path.join(%(tmp)r, "ok%%d" %% rank), "w").write(repr(report))
dist.destroy_process_group()
'''


def test_row_partitioned_renderer_classes_match_the_single_gpu_renderer(tmp_path):
    """Two ranks (gloo, one GPU): `SurfaceSplattingRenderer(row_partition=...)` -- full-image output with a replicated loss and
    band output with `band_image_loss`, owner and bucket gradient exchange, contiguous and tile-row-cyclic bands -- against the
    plain renderer: image bit for bit, `points.grad` / `colors.grad` <= 1e-5."""
    script = os.path.join(str(tmp_path), "two_rank_classes.py")
    open(script, "w").write(_TWO_RANK % {"root": ROOT, "tmp": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29721", script],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-12000:]
    assert os.path.exists(os.path.join(str(tmp_path), "ok0")) and os.path.exists(os.path.join(str(tmp_path), "ok1"))


_TWO_RANK_EDGE = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import scenes
from dss_amd.cameras import FoVPerspectiveCameras, look_at_view_transform
from dss_amd.cloud import PointClouds3D, PointCloudsFilters
from dss_amd.distributed import RowPartition
from dss_amd.rasterizer import PointsRasterizationSettings, SurfaceSplatting
from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("gloo")
pts, nrm = scenes.load_cloud("bunny")
pts = scenes.normalize_unit_sphere(pts)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
gen = torch.Generator().manual_seed(7)


def settings(S, **kw):
    d = dict(backface_culling=False, cutoff_threshold=1.0, depth_merging_threshold=0.05, Vrk_invariant=True, Vrk_isotropic=False,
             radii_backward_scaler=5.0, image_size=S, points_per_pixel=5, bin_size=None, clip_pts_grad=0.05, antialiasing_sigma=1.0)
    d.update(kw)
    return PointsRasterizationSettings(**d)


def case(name, S, cams, clouds_fn, st, parts, verbose=False, with_filter=False, **ren_kw):
    """plain renderer vs the row-partitioned one: image bit for bit, gradients of a fixed linear functional of the image"""
    out = {}
    for key, part in (("single", None),) + tuple(("part%%d" %% i, p) for i, p in enumerate(parts)):
        Xs, Cs, cloud = clouds_fn()
        ras = SurfaceSplatting(cameras=cams, raster_settings=st)
        kw = dict(ren_kw)
        if part is not None:
            kw["row_partition"] = part
            if isinstance(part, str):
                from dss_amd.sharded import default_partition
                part = RowPartition(S, world, rank)      # what "auto" must fall back to here (five channels)
        ren = SurfaceSplattingRenderer(ras, NormWeightedCompositor(), **kw)
        flt = PointCloudsFilters(device=dev, activation=torch.ones((len(Xs), Xs[0].shape[0]), dtype=torch.bool, device=dev)) if with_filter else None
        res = ren(cloud, verbose=verbose, point_clouds_filter=flt) if with_filter else ren(cloud, verbose=verbose)
        img, frag = res if verbose else (res, None)
        if key == "single":
            w = torch.randn(img.shape, generator=gen).to(dev)
            out["w"] = w
        (img * out["w"]).sum().backward()
        out[key] = (img.detach(), [x.grad.clone() for x in Xs], [c.grad.clone() for c in Cs], frag,
                    None if flt is None else flt.visibility.clone())
    ref = out["single"]
    for i, part in enumerate(parts):
        part = RowPartition(S, world, rank) if isinstance(part, str) else part
        got = out["part%%d" %% i]
        assert torch.equal(got[0], ref[0]), (name, "image differs", i)
        for a, b in zip(got[1] + got[2], ref[1] + ref[2]):
            assert rel(a, b) < 1e-5, (name, i, rel(a, b))
        if verbose:   # the fragments of a partitioned render are those of the rank's rows
            ri = torch.tensor(part.row_indices(), device=dev, dtype=torch.int64)
            assert torch.equal(got[3].idx, ref[3].idx.index_select(1, ri)) and torch.equal(got[3].zbuf, ref[3].zbuf.index_select(1, ri))
        if with_filter:   # the visibility handed to the filter object is the union over the ranks
            assert torch.equal(got[4], ref[4]), (name, "visibility filter differs")
    return out


# (1) N clouds for N cameras (not shared), five feature channels, fragments requested, the filter object's visibility
S = 128
R, T = look_at_view_transform(2.0, 25.0, [40.0, 160.0, 280.0])
cams3 = FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R, T=T, device=dev)


def three_clouds():
    Xs = [torch.nn.Parameter(t(pts[i::3]).clone() * (1.0 + 0.05 * i)) for i in range(3)]
    Cs = [torch.nn.Parameter(torch.rand((x.shape[0], 5), generator=torch.Generator().manual_seed(i)).to(dev)) for i, x in enumerate(Xs)]
    return Xs, Cs, PointClouds3D(Xs, [t(nrm[i::3]) for i in range(3)], Cs)


case("per-camera clouds, C=5", S, cams3, three_clouds, settings(S),
     [RowPartition(S, world, rank), RowPartition(S, world, rank, bounds=[0, 48, S])], verbose=True, with_filter=True)
# (the tile-row-cyclic variants of the backward are built for RGB features: an explicit cyclic partition with five channels is
# refused with a clear message, "auto" falls back to contiguous bands)
try:
    Xs, Cs, cloud = three_clouds()
    SurfaceSplattingRenderer(SurfaceSplatting(cameras=cams3, raster_settings=settings(S)), NormWeightedCompositor(),
                             row_partition=RowPartition(S, world, rank, cyclic=True))(cloud)
    raise AssertionError("a cyclic partition with five feature channels must be refused")
except ValueError as e:
    assert "3 feature channels" in str(e)
case("per-camera clouds, C=5, auto", S, cams3, three_clouds, settings(S), ["auto"])

# (2) one camera, unequal contiguous bands incl. a rank WITHOUT rows, isotropic per-point scale
cam1 = FoVPerspectiveCameras(znear=0.1, zfar=100.0, fov=60.0, R=R[:1], T=T[:1], device=dev)


def one_cloud():
    X = torch.nn.Parameter(t(pts).clone())
    C = torch.nn.Parameter(torch.rand((pts.shape[0], 3), generator=torch.Generator().manual_seed(9)).to(dev))
    return [X], [C], PointClouds3D([X], [t(nrm)], [C])


S2 = 96
case("one camera, bands (0, 96, 96) / (0, 40, 96), isotropic", S2, cam1, one_cloud, settings(S2, Vrk_invariant=False, Vrk_isotropic=True),
     [RowPartition(S2, world, rank, bounds=[0, 96, 96]), RowPartition(S2, world, rank, bounds=[0, 40, 96])], gradient_exchange="owner")
case("one camera, bucket", S2, cam1, one_cloud, settings(S2), [RowPartition(S2, world, rank, bounds=[0, 56, 96])], gradient_exchange="bucket")

# (3) cameras that cull different points of one shared cloud (per-camera variance scale), band output
R2, T2 = look_at_view_transform([2.0, 2.2], [20.0, -10.0], [30.0, 200.0])
cams2 = FoVPerspectiveCameras(fov=60.0, R=R2, T=T2, device=dev)
cams2.znear = torch.tensor([1.9, 1.0], device=dev)
cams2.zfar = torch.tensor([100.0, 2.3], device=dev)
case("shared cloud, different culling per camera", S, cams2, one_cloud, settings(S), [RowPartition(S, world, rank, cyclic=True)])
open(os.path.join(%(tmp)r, "edge_ok%%d" %% rank), "w").write("ok")
dist.destroy_process_group()
'''


def test_row_partitioned_renderer_edge_cases(tmp_path):
    """Two gloo ranks on one GPU through the classes: per-camera clouds with five feature channels, fragments (`verbose`) and
    the filter object's visibility; one camera with unequal bands and a rank WITHOUT rows, isotropic scale; the bucket form;
    cameras that cull different points of a shared cloud."""
    script = os.path.join(str(tmp_path), "two_rank_edge.py")
    open(script, "w").write(_TWO_RANK_EDGE % {"root": ROOT, "tmp": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29723", script],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-12000:]
    assert os.path.exists(os.path.join(str(tmp_path), "edge_ok0")) and os.path.exists(os.path.join(str(tmp_path), "edge_ok1"))
