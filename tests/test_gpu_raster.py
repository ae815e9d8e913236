"""GPU parity tests of the HIP rasterizer (through the C ABI via dss_amd.ops) against
 (a) the committed outputs of the unmodified reference CPU rasterizer (tests/golden/ref_*.npz),
 (b) the oracle on seeded inputs, and (c) size-independent properties at BASELINE sizes.
Integer / index outputs and the fp32 fragment values must be BIT-EXACT; accumulated gradients
are compared with rel-L2 <= 1e-3 (north_star tolerance; observed ~1e-6)."""
import os

import numpy as np
import pytest
import torch

import oracle
import scenes
from dss_amd import _lib, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["ref_teapot256", "ref_random48", "ref_random64x2", "ref_ties32"]


def _dev(sc):
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(DEV)
    return dict(points=t("points"), ellipse=t("ellipse"), cutoff=t("cutoff"), radii=t("radii"),
                first=t("first_idx"), num=t("num_pts"))


def _fwd(d, S, K, thr, bin_size=None, **kw):
    return ops.splat_points(d["points"], d["ellipse"], d["cutoff"], d["radii"], d["first"], d["num"], thr, S, K,
                            bin_size, None, **kw)


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("bin_size", [None, 0])
@pytest.mark.parametrize("name", CASES)
def test_forward_bit_exact_vs_reference_golden(golden_dir, name, bin_size):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    S, K, thr = int(z["S"]), int(z["K"]), float(z["thr"])
    idx, zbuf, qv, occ = _fwd(_dev(z), S, K, thr, bin_size)
    assert idx.dtype == torch.int32 and tuple(idx.shape) == z["ref_idx"].shape
    assert np.array_equal(idx.cpu().numpy(), z["ref_idx"])
    assert np.array_equal(zbuf.cpu().numpy(), z["ref_zbuf"])
    assert np.array_equal(qv.cpu().numpy(), z["ref_qvalue"])
    assert np.array_equal(occ.cpu().numpy(), z["ref_occ"])


@pytest.mark.parametrize("K", [1, 2, 5, 8, 11, 16, 20, 32, 33, 64, 150])
def test_forward_all_k_vs_oracle(K):
    sc = scenes.random_splats(900, 56, 2, seed=K, rmin=2.0, rmax=9.0)
    S, thr = 56, 0.4
    got = _fwd(_dev(sc), S, K, thr)
    want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                sc["num_pts"], S, K, thr)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)


@pytest.mark.parametrize("K,thr,ties", [(5, 0.05, False), (5, 0.4, True), (1, 0.05, False), (8, 10.0, False), (12, 0.05, True)])
@pytest.mark.parametrize("bin_size", [None, 0])
def test_forward_deep_overlap_with_the_footprint_depth_cut(K, thr, ties, bin_size):
    """Hundreds of overlapping splats per pixel (the state the reference's training loop reaches at configs[2]): tiles take
    many candidate chunks, and from the second chunk on the fine kernel drops candidates behind every pixel's K-th entry or
    farther than the merge threshold behind every pixel's nearest entry before their ellipse tests (raster_forward.hip,
    `cut_k` / `cut_0`).  Fragments stay bit-exact, with depth ties ((z, idx) order) and with a threshold that never cuts."""
    S = 64
    sc = scenes.random_splats(5000, S, 2, seed=K + int(thr * 100), rmin=3.0, rmax=8.0, ties=ties)
    got = _fwd(_dev(sc), S, K, thr, bin_size)
    want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                sc["num_pts"], S, K, thr)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)


def test_forward_k_too_large_raises():
    sc = scenes.random_splats(10, 16, 1)
    with pytest.raises(RuntimeError, match="kMaxPointsPerPixel"):
        _fwd(_dev(sc), 16, 151, 0.05)


@pytest.mark.parametrize("S", [1, 7, 16, 17, 33, 100])
def test_forward_ragged_image_sizes(S):
    sc = scenes.random_splats(300, S, 2, seed=S, rmin=0.6, rmax=5.0)
    got = _fwd(_dev(sc), S, 5, 0.05)
    want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                sc["num_pts"], S, 5, 0.05)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)


def test_forward_empty_and_ragged_clouds():
    # cloud 0 empty, cloud 1 has 5 points, cloud 2 has 300; plus a gap of unowned points
    sc = scenes.random_splats(400, 32, 1, seed=3)
    sc["first_idx"] = np.array([0, 10, 100], np.int64)
    sc["num_pts"] = np.array([0, 5, 300], np.int64)
    for bs in (None, 0):
        got = _fwd(_dev(sc), 32, 5, 0.05, bs)
        want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                    sc["num_pts"], 32, 5, 0.05)
        for g, w in zip(got, want):
            assert np.array_equal(g.cpu().numpy(), w)
    assert got[3][0].sum().item() == 0
    # P == 0
    e = lambda *s: torch.zeros(*s, device=DEV)
    idx, zbuf, qv, occ = ops.splat_points(e(0, 3), e(0, 3), e(0), e(0, 2), torch.zeros(1, dtype=torch.int64, device=DEV),
                                          torch.zeros(1, dtype=torch.int64, device=DEV), 0.05, 20, 5)
    assert (idx == -1).all() and (zbuf == -1).all() and (qv == -1).all() and (occ == 0).all()


def test_forward_list_overflow_goes_through_the_spill_pool():
    """Splats far larger than a tile put far more entries on every tile than the fixed-capacity sub-lists hold (600
    splats x 15-60 px radius on 256 tiles): the overflowed sub-lists are re-binned into the spill pool -- and, since
    there are more such pairs than the pool has entries, part of the tiles fall back to whole-cloud scans.  Exact."""
    sc = scenes.random_splats(600, 128, 1, seed=4, rmin=30.0, rmax=60.0)
    got = _fwd(_dev(sc), 128, 5, 10.0)
    want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                sc["num_pts"], 128, 5, 10.0)
    for g, w in zip(got, want):
        assert np.array_equal(g.cpu().numpy(), w)


@pytest.mark.parametrize("rmin,rmax,P", [(9.0, 26.0, 6000), (2.0, 45.0, 8000), (30.0, 70.0, 3000)])
def test_forward_overflowed_lists_of_wide_splats_stay_on_the_lists(rmin, rmax, P):
    """Sub-lists far over capacity, filled by splats wider than 2 x 2 tiles: up to 8 x 8 tiles a splat records its full tiles in
    a 64-bit mask of its own, wider ones append one record per 8 x 8 block (raster_forward.hip, `Spill::big` / `giant`), and
    the pool pass moves all of them into the pool -- the state of the reference's training loop at configs[2] (10-pixel splats,
    hundreds per pixel), where round 5 scanned the whole cloud for every spilled tile.  Exact, two clouds, both bin modes."""
    S = 128
    sc = scenes.random_splats(P, S, 2, seed=int(rmax), rmin=rmin, rmax=rmax)
    want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                sc["num_pts"], S, 5, 0.05)
    for bs in (None, 0):
        got = _fwd(_dev(sc), S, 5, 0.05, bs)
        for g, w in zip(got, want):
            assert np.array_equal(g.cpu().numpy(), w)


def test_lean_workspace_mode_is_exact_and_smaller():
    """dss_set_option(DSS_OPT_LEAN_WORKSPACE, 1) halves the sub-list capacity and drops the packed records (raster_forward.hip
    `lean_workspace`): smaller workspace, same fragments bit for bit, same image."""
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, 4, seed=0)
    h = scenes.global_h(pts)
    S, K, thr = 256, 5, 0.05
    M, V, _ = scenes.camera_matrices(2.0, 20.0, 30.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    P = pts.shape[0]
    a = (t(pts), t(nrm), torch.full((1,), h, device=DEV), t(M), t(V), torch.full((1,), 0.1, device=DEV),
         torch.full((1,), 100.0, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV),
         torch.full((1,), P, dtype=torch.int64, device=DEV), torch.rand((P, 3), device=DEV))
    lib = _lib.load()
    full_bytes = lib.dss_render_forward_workspace(1, 4_000_000, 2048, K)
    ref = ops.render_forward(*a, S, K, 1.0, thr, 1.0, False, True, workspace_state=0)
    _lib.set_option(_lib.OPT_LEAN_WORKSPACE, 1)
    try:
        lean_bytes = lib.dss_render_forward_workspace(1, 4_000_000, 2048, K)
        got = ops.render_forward(*a, S, K, 1.0, thr, 1.0, False, True, workspace_state=0)
    finally:
        _lib.set_option(_lib.OPT_LEAN_WORKSPACE, 0)
    assert lean_bytes < 0.3 * full_bytes and lean_bytes <= 128 << 20, (lean_bytes, full_bytes)  # configs[4]: 433 -> 110 MB
    for k in ("idx", "zbuf", "qvalue", "occupancy", "visible"):
        assert torch.equal(ref[k], got[k]), k
    assert float((ref["image"] - got["image"]).abs().max()) <= 1e-6


@pytest.mark.parametrize("fused", [False, True])
def test_far_camera_overflow_is_exact_and_not_a_cliff(fused):
    """A camera far away concentrates the whole cloud on a few tiles: ~100k points on 1/30 of a 512^2 screen put
    hundreds of entries on sub-lists sized for a mean load of 6.  Round 1 rasterized every such tile from its whole cloud
    (O(64 P) per tile: 30x the step time); with the spill pool the result is still bit-exact against the oracle and the
    forward stays within 2x of the near-camera forward of the same cloud (measured: 1.7x at 1/30 of the screen, 2.2x at
    1/70, 4.3x at 1/220 -- the remaining cost is the concentration itself: few tiles, long lists, hot counters)."""
    import time
    pts, nrm = scenes.load_cloud("yoga6")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, 10, seed=0)
    h = scenes.global_h(pts)
    S, K, thr = 512, 5, 0.05
    times = {}
    for tag, dist in (("near", 2.0), ("far", 4.0)):
        M, V, _ = scenes.camera_matrices(dist, 20.0, 30.0)
        sc = scenes.setup_scene(pts, nrm, M, V, S, h=h)
        d = _dev(sc)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        if fused:
            a = (t(pts), t(nrm), torch.full((1,), h, device=DEV), t(M), t(V), torch.full((1,), 0.1, device=DEV),
                 torch.full((1,), 100.0, device=DEV), t(sc["first_idx"]), t(sc["num_pts"]),
                 torch.ones((pts.shape[0], 3), device=DEV))
            run = lambda: ops.render_forward(*a, S, K, 1.0, thr, 1.0, False, True)
            got = run()
            got = (got["idx"], got["zbuf"], got["qvalue"], got["occupancy"])
        else:
            run = lambda: _fwd(d, S, K, thr)
            got = run()
        want = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"], sc["num_pts"],
                                    S, K, thr)
        for g_, w_ in zip(got, want):
            assert np.array_equal(g_.cpu().numpy(), w_), tag
        if tag == "far":
            assert float(want[3].mean()) < 1.0 / 16, "the far view is meant to cover a small part of the screen"
        best = 1e9
        for _ in range(5):   # best of five batches: the box is shared with the rest of the suite's allocations
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        times[tag] = best
    assert times["far"] <= 2.0 * times["near"], times


@pytest.mark.parametrize("dist", [2.0, 5.0])
def test_cell_ordered_binning_of_large_inputs_is_exact(dist):
    """More than 2,000,000 splats take the cell-ordered binning of dss_render_forward (setup + cell histogram -> scans ->
    scatter -> LDS-aggregated binning -> queue build, raster_forward.hip `bin_sorted_kernel`); the direct binning of
    `splat_points` fills the same lists in another order.  Same fragments bit for bit -- near camera (primary lists), far camera (the whole
    cloud on a few tiles: the sub-lists overflow and the spill pool is filled by the sorted path's mask bytes) and a
    second cloud with a gap of unowned points."""
    pts, nrm = scenes.load_cloud("yoga6")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, 106, seed=0)     # 1,057,774 points per cloud, two clouds
    h = scenes.global_h(pts[::40]) / 40.0
    S, K, thr = 512, 5, 0.05
    M = np.concatenate([scenes.camera_matrices(dist, 20.0, a)[0] for a in (30.0, 170.0)])
    V = np.concatenate([scenes.camera_matrices(dist, 20.0, a)[1] for a in (30.0, 170.0)])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    Pc = len(pts)
    first = torch.tensor([0, Pc], dtype=torch.int64, device=DEV)
    num = torch.tensor([Pc, Pc - 1000], dtype=torch.int64, device=DEV)     # the last 1000 packed points belong to no cloud
    f = ops.render_forward(t(pts), t(nrm), torch.full((2,), float(h), device=DEV), t(M), t(V), torch.full((2,), 0.1, device=DEV),
                           torch.full((2,), 100.0, device=DEV), first, num, torch.rand((2 * Pc, 3), device=DEV), S, K, 1.0, thr,
                           1.0, False, True)
    want = ops.splat_points(f["pts_screen"], f["ellipse_params"], f["cutoff_threshold"], f["radii"], first, num, thr, S, K,
                            return_visible=True)
    assert float(want[3].mean()) > 0.005
    if dist > 4.0:
        assert float(want[3].mean()) < 1.0 / 16, "the far view is meant to cover a small part of the screen"
    for k, w_ in zip(("idx", "zbuf", "qvalue", "occupancy", "visible"), want):
        assert torch.equal(f[k], w_), (k, dist)
    # the workspace is left clean by the sorted path as well: a second call gives the same result
    f2 = ops.render_forward(t(pts), t(nrm), torch.full((2,), float(h), device=DEV), t(M), t(V), torch.full((2,), 0.1, device=DEV),
                            torch.full((2,), 100.0, device=DEV), first, num, torch.rand((2 * Pc, 3), device=DEV), S, K, 1.0, thr,
                            1.0, False, True)
    assert torch.equal(f2["idx"], f["idx"]) and torch.equal(f2["occupancy"], f["occupancy"])
    # row bands (multi-GPU) on the same path: a contiguous band and a tile-row-cyclic one are rows of the full render
    from dss_amd.distributed import RowPartition
    feat = torch.rand((2 * Pc, 3), device=DEV)
    args = (t(pts), t(nrm), torch.full((2,), float(h), device=DEV), t(M), t(V), torch.full((2,), 0.1, device=DEV),
            torch.full((2,), 100.0, device=DEV), first, num, feat, S, K, 1.0, thr, 1.0, False, True)
    for part in (RowPartition(S, 4, 1), RowPartition(S, 4, 2, cyclic=True)):
        fb = ops.render_forward(*args, rows=part.rows)
        own = torch.tensor(part.row_indices(), device=DEV, dtype=torch.int64)
        assert torch.equal(fb["idx"], f["idx"][:, own]) and torch.equal(fb["qvalue"], f["qvalue"][:, own]), part.describe()


def test_cell_ordered_binning_keeps_overflowing_wide_splats_on_the_lists():
    """The cell-ordered path (> 2M splats) with WIDE splats among them: 1 % of the points get a variance scale 4000x the cloud's
    (20-40 pixel splats, part of them wider than 8 x 8 tiles), sub-lists overflow under them, and `bin_sorted_kernel` records their full
    tiles like the direct binning does (64-bit masks / block records: raster_forward.hip `Spill::big`, `giant`).  Same
    fragments as the direct binning of `splat_points`, bit for bit, twice (the workspace is left clean)."""
    pts, nrm = scenes.load_cloud("yoga6")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, 106, seed=0)     # 1,057,774 points per cloud, two clouds
    h0 = scenes.global_h(pts[::40]) / 40.0
    S, K, thr = 512, 5, 0.05
    # cloud 0 seen from nearby, cloud 1 from far away: there the whole cloud lands on a few tiles and every sub-list overflows
    M = np.concatenate([scenes.camera_matrices(d, 20.0, a)[0] for d, a in ((2.0, 30.0), (5.0, 170.0))])
    V = np.concatenate([scenes.camera_matrices(d, 20.0, a)[1] for d, a in ((2.0, 30.0), (5.0, 170.0))])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    Pc = len(pts)
    first = torch.tensor([0, Pc], dtype=torch.int64, device=DEV)
    num = torch.tensor([Pc, Pc], dtype=torch.int64, device=DEV)
    h = torch.full((2 * Pc,), float(h0), device=DEV)
    pick = torch.from_numpy(np.random.default_rng(0).permutation(2 * Pc)).to(DEV)
    h[pick[:(2 * Pc) // 100]] = float(h0) * 4000.0      # 20-40 pixels near, ~10 far
    h[pick[:(2 * Pc) // 1000]] = float(h0) * 60000.0    # wider than 8 x 8 tiles in the far view as well
    world, normals = t(np.concatenate([pts, pts])), t(np.concatenate([nrm, nrm]))
    args = (world, normals, h, t(M), t(V), torch.full((2,), 0.1, device=DEV), torch.full((2,), 100.0, device=DEV), first, num,
            torch.rand((2 * Pc, 3), device=DEV), S, K, 1.0, thr, 1.0, False, False)
    f = ops.render_forward(*args)
    rad_px = f["radii"].amax(1) * (S / 2)
    far = rad_px[Pc:]
    assert int((far > 8).sum()) > 3000 and int((far > 28).sum()) > 100, (int((far > 8).sum()), int((far > 28).sum()))
    assert float(f["occupancy"][1].mean()) < 0.5
    want = ops.splat_points(f["pts_screen"], f["ellipse_params"], f["cutoff_threshold"], f["radii"], first, num, thr, S, K,
                            return_visible=True)
    for k, w_ in zip(("idx", "zbuf", "qvalue", "occupancy", "visible"), want):
        assert torch.equal(f[k], w_), k
    f2 = ops.render_forward(*args)
    assert torch.equal(f2["idx"], f["idx"]) and torch.equal(f2["occupancy"], f["occupancy"])


@pytest.mark.parametrize("reps,S,form,C", [(4, 256, 0, 3), (4, 256, 1, 3), (16, 512, 0, 3), (4, 192, 0, 3), (4, 192, 0, 5), (16, 320, 0, 5)])
def test_owner_mode_of_the_band_backward(reps, S, form, C):
    """dss_render_backward_owned (`render_backward(grad_out_full=...)`): on a row band the occupancy surrogate of a (camera,
    point) pair is computed -- whole window, full image gradient -- by the ONE rank whose band holds the image row of the
    point's centre.  Over the ranks of a partition (contiguous, unequal, tile-row-cyclic): every pair has a non-zero position
    gradient on at most one rank, the ranks' position gradients sum to the whole-image backward, and so do the partial
    feature gradients.  Short lists in the two-launch form with the filter inside the gather (form 0) and in the round-3
    launch sequence (form 1), a list above 262,144 points (cell-sorted gather), image sizes that are not powers of two (the
    rows' NDC then comes from the reference expression, not from exact additions) and five feature channels (generic-channel
    kernels; a tile-row-cyclic band is built for RGB only and is left out there)."""
    from dss_amd import _lib
    from dss_amd.distributed import RowPartition
    pts, nrm = scenes.load_cloud("yoga6")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, reps, seed=0)
    Pc = len(pts)
    h = scenes.global_h(pts[:: max(1, Pc // 25000)]) * (25000.0 / Pc if Pc > 25000 else 1.0)
    K, thr, N = 5, 0.05, 2
    M = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[0] for a in (30.0, 170.0)])
    V = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[1] for a in (30.0, 170.0)])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.tensor([0, Pc], dtype=torch.int64, device=DEV)
    num = torch.tensor([Pc, Pc - 100], dtype=torch.int64, device=DEV)
    feat = torch.rand((N * Pc, C), device=DEV)
    args = (t(pts), t(nrm), torch.full((N,), float(h), device=DEV), t(M), t(V), torch.full((N,), 0.1, device=DEV),
            torch.full((N,), 100.0, device=DEV), first, num, feat, S, K, 1.0, thr, 1.0, False, True)
    full = ops.render_forward(*args)
    go = torch.randn((N, S, S, C + 1), device=DEV)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    try:
        if form:
            _lib.set_option(_lib.OPT_BACKWARD_FUSED, form)
        gf0, gp0 = ops.render_backward(go, full["idx"], full["qvalue"], full["wsum"], full["scaler"], full["pts_screen"],
                                       full["radii"], full["visible"], first, num, 4.0, -1.0)
        assert float(gp0.abs().sum()) > 0 and float(gf0.abs().sum()) > 0
        layouts = [[RowPartition(S, 4, r) for r in range(4)], [RowPartition(S, 3, r, bounds=[0, 8, S - 40, S]) for r in range(3)]]
        if C == 3:
            layouts.append([RowPartition(S, 4, r, cyclic=True) for r in range(4)])
        for parts in layouts:
            gf_sum, gp_sum, owners = torch.zeros_like(gf0), torch.zeros_like(gp0), torch.zeros(N * Pc, device=DEV)
            for part in parts:
                own = torch.tensor(part.row_indices(), device=DEV, dtype=torch.int64)
                o = ops.render_forward(*args, rows=part.rows)
                gf, gp = ops.render_backward(go[:, own].contiguous(), o["idx"], o["qvalue"], o["wsum"], o["scaler"], o["pts_screen"],
                                             o["radii"], full["visible"], first, num, 4.0, -1.0, image_size=S, rows=part.rows,
                                             grad_out_full=go)
                gf_sum += gf
                gp_sum += gp
                owners += (gp != 0).any(dim=1).float()
            assert float(owners.max()) <= 1.0, parts[0].describe()
            # every pair the whole-image backward moves has an owner (its centre lies on the image)
            assert bool((~(gp0 != 0).any(dim=1) | (owners > 0)).all()), parts[0].describe()
            assert rel(gp_sum, gp0) < 1e-5 and rel(gf_sum, gf0) < 1e-5, (parts[0].describe(), rel(gp_sum, gp0), rel(gf_sum, gf0))
    finally:
        _lib.set_option(_lib.OPT_BACKWARD_FUSED, 0)


@pytest.mark.parametrize("reps,S", [(1, 128), (4, 256), (106, 512)])
def test_band_outputs_only_gives_the_same_band_and_the_same_gradients(reps, S):
    """DSS_WS_BAND_OUTPUTS (multi-GPU ranks; `render_forward(band_outputs_only=True)`): the splats that miss the rank's rows
    keep out of the sort / binning and write position, radii and validity only.  The band's fragments, image, weight sums and
    visibility flags are the same bits as without the flag; position / radii / validity are complete; ellipse / scaler /
    cutoff are those of the plain call wherever a splat is visible; and the backward through these outputs (global
    visibility) gives the same partial gradients.  Direct binning (small clouds) and the cell-ordered path of more than 2M
    splats in all three forms: sorting for itself, saving the point order, reusing it."""
    from dss_amd.distributed import RowPartition
    pts, nrm = scenes.load_cloud("yoga6")
    pts = scenes.normalize_unit_sphere(pts)
    if reps > 1:
        pts, nrm = scenes.upsample_jitter(pts, nrm, reps, seed=0)
    Pc = len(pts)
    h = scenes.global_h(pts[:: max(1, Pc // 25000)]) * (25000.0 / Pc if Pc > 25000 else 1.0)
    K, thr, N = 5, 0.05, 2
    M = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[0] for a in (30.0, 170.0)])
    V = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[1] for a in (30.0, 170.0)])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.tensor([0, Pc], dtype=torch.int64, device=DEV)
    num = torch.tensor([Pc, Pc - 100], dtype=torch.int64, device=DEV)     # the last 100 packed points belong to no cloud
    feat = torch.rand((N * Pc, 3), device=DEV)
    args = (t(pts), t(nrm), torch.full((N,), float(h), device=DEV), t(M), t(V), torch.full((N,), 0.1, device=DEV),
            torch.full((N,), 100.0, device=DEV), first, num, feat, S, K, 1.0, thr, 1.0, False, True)
    full = ops.render_forward(*args)
    assert float(full["occupancy"].mean()) > 0.01
    go = torch.randn((N, S, S, 4), device=DEV)
    large = N * Pc > 2_000_000
    for part in (RowPartition(S, 4, 1), RowPartition(S, 4, 2, cyclic=True), RowPartition(S, 4, 3, bounds=[0, 8, 40, S - 24, S])):
        own = torch.tensor(part.row_indices(), device=DEV, dtype=torch.int64)
        plain = ops.render_forward(*args, rows=part.rows)
        kinds = [dict(order_refresh=0)] + ([dict(order_refresh=3), dict(order_refresh=3)] if large else [])   # sort | save | reuse
        for kw in kinds:
            f = ops.render_forward(*args, rows=part.rows, band_outputs_only=True, **kw)
            for k in ("idx", "zbuf", "qvalue", "occupancy", "image", "wsum", "visible", "pts_screen", "radii", "valid"):
                assert torch.equal(f[k], plain[k]), (k, part.describe(), kw)
            vis = plain["visible"]
            for k in ("ellipse_params", "scaler", "cutoff_threshold"):
                assert torch.equal(f[k][vis], plain[k][vis]), (k, part.describe(), kw)
            bw = lambda o: ops.render_backward(go[:, own].contiguous(), o["idx"], o["qvalue"], o["wsum"], o["scaler"], o["pts_screen"],
                                               o["radii"], full["visible"], first, num, 4.0, -1.0, image_size=S, rows=part.rows)
            (gf, gp), (gf0, gp0) = bw(f), bw(plain)
            assert torch.equal(gf, gf0) and torch.equal(gp, gp0), (part.describe(), kw)
    if large:
        # ONE saved order reused by different bands in turn (the workspace is shared): under DSS_WS_BAND_OUTPUTS only the splats
        # that meet the band store their candidate record, so the records of the previous band's splats are stale where
        # the next band does not reach -- they must never be binned (position bytes, setup_cell_kernel / bin_sorted_kernel)
        seq = [(RowPartition(S, 4, 1), True), (RowPartition(S, 4, 1), True), (RowPartition(S, 4, 3), True),
               (RowPartition(S, 1, 0), False), (RowPartition(S, 4, 0), True), (RowPartition(S, 4, 2, cyclic=True), True)]
        ops.render_forward(*args, rows=seq[0][0].rows, order_refresh=0)     # (forget the order of the loop above)
        for i, (part, band_only) in enumerate(seq):                         # call 0 saves, calls 1.. reuse
            f = ops.render_forward(*args, rows=part.rows, band_outputs_only=band_only, order_refresh=len(seq) + 1)
            plain = ops.render_forward(*args, rows=part.rows, workspace_state=0)
            for k in ("idx", "zbuf", "qvalue", "occupancy", "image", "wsum", "visible"):
                assert torch.equal(f[k], plain[k]), (k, i, part.describe())


def test_row_bands_concatenate_to_full_image():
    sc = scenes.random_splats(2000, 96, 2, seed=6)
    d = _dev(sc)
    full = _fwd(d, 96, 5, 0.05, return_visible=True)
    for bounds in ([0, 48, 96], [0, 12, 24, 36, 48, 60, 72, 84, 96], [0, 5, 50, 96]):
        parts = [_fwd(d, 96, 5, 0.05, rows=(a, b), return_visible=True) for a, b in zip(bounds[:-1], bounds[1:])]
        for i in range(4):
            assert torch.equal(torch.cat([p[i] for p in parts], dim=1), full[i])
        vis = torch.stack([p[4] for p in parts]).any(0)
        assert torch.equal(vis, full[4])


def _cfg2_scene():
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    pts, nrm = scenes.upsample_jitter(pts, nrm, 4, seed=0)
    M, V, _ = scenes.camera_matrices(2.0, 30.0, 45.0)
    rng = np.random.default_rng(0)
    return scenes.setup_scene(pts, nrm, M, V, 512, colors=rng.uniform(0, 1, (pts.shape[0], 3)).astype(np.float32))


def test_cfg2_bunny_512_forward_backward_vs_oracle():
    """BASELINE config 2: bunny x4 (32,684 pts), 1 camera, 512^2, K=5, fwd+bwd, checked against the oracle."""
    sc = _cfg2_scene()
    S, K, thr, radii_s, clip = 512, 5, 0.05, 5.0, 0.05
    P = sc["points"].shape[0]
    assert P == 32684
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, S, K, thr, return_visible=True)
    o_idx, o_zbuf, o_qv, o_occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"],
                                                      sc["first_idx"], sc["num_pts"], S, K, thr)
    assert np.array_equal(idx.cpu().numpy(), o_idx) and np.array_equal(zbuf.cpu().numpy(), o_zbuf)
    assert np.array_equal(qv.cpu().numpy(), o_qv) and np.array_equal(occ.cpu().numpy(), o_occ)
    assert np.array_equal(vis.cpu().numpy(), oracle.visibility(o_idx, P))

    # blend forward: <= 1e-4 RGB
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    feat = torch.from_numpy(sc["colors"]).to(DEV)
    img = ops.blend_forward(idx, qv, occ, scaler, feat)
    o_img = oracle.blend_forward(o_idx, o_qv, o_occ, sc["scaler"], sc["colors"])
    assert np.abs(img.cpu().numpy() - o_img).max() <= 1e-4

    # backward with grad_out = randn(seed 1) on RGBA
    rng = np.random.default_rng(1)
    grad_out = rng.standard_normal((1, S, S, 4)).astype(np.float32)
    o_gf, o_gocc = oracle.blend_backward(grad_out, o_idx, o_qv, sc["scaler"], P)
    img2, wsum = ops.blend_forward(idx, qv, occ, scaler, feat, return_wsum=True)
    assert torch.equal(img2, img)
    geom = (d["points"], d["radii"], vis, d["first"], d["num"])
    for kw in (dict(), dict(geometry=geom), dict(geometry=geom, wsum=wsum)):  # scatter / gather / gather+wsum
        gf, gocc = ops.blend_backward(torch.from_numpy(grad_out).to(DEV), idx, qv, scaler, P, **kw)
        assert np.array_equal(gocc.cpu().numpy(), o_gocc)
        assert _rel_l2(gf.cpu().numpy(), o_gf) <= 1e-3, kw.keys()
    gf2, _ = ops.blend_backward(torch.from_numpy(grad_out).to(DEV), idx, qv, scaler, P, geometry=geom, wsum=wsum)
    assert torch.equal(gf, gf2)  # gather formulation is deterministic

    grad_zbuf = rng.standard_normal((1, S, S, K)).astype(np.float32)
    for gz, cl in ((None, clip), (grad_zbuf, -1.0), (grad_zbuf, clip)):
        g, rs = ops.splat_backward(d["points"], d["radii"], vis, idx, gocc,
                                   None if gz is None else torch.from_numpy(gz).to(DEV), d["first"], d["num"],
                                   radii_s, cl, return_rs=True)
        o_g, o_vis, o_rs = oracle.splat_backward(sc["points"], sc["radii"], o_idx, o_gocc, gz, sc["first_idx"],
                                                 sc["num_pts"], radii_s, cl)
        assert np.array_equal(rs.cpu().numpy(), o_rs)
        assert _rel_l2(g.cpu().numpy(), o_g) <= 1e-3, (cl, gz is None)
        assert np.isfinite(g.cpu().numpy()).all()

    # fused single-GPU backward (persistent wavefronts over the compacted visible list, four points per wavefront): the
    # same per-pair terms as the one-wavefront-per-point kernels, summed in a different fixed order
    def same(a, b, tol=2e-6):
        return _rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= tol
    g_ref, rs_ref = ops.splat_backward(d["points"], d["radii"], vis, idx, gocc, None, d["first"], d["num"], radii_s,
                                       clip, return_rs=True)
    gf_ref, _ = ops.blend_backward(torch.from_numpy(grad_out).to(DEV), idx, qv, scaler, P, geometry=geom, wsum=wsum)
    for ws_ in (wsum, None):
        gf_f, g_f, rs_f = ops.render_backward(torch.from_numpy(grad_out).to(DEV), idx, qv, ws_, scaler, d["points"],
                                              d["radii"], vis, d["first"], d["num"], radii_s, clip, return_rs=True)
        assert torch.equal(rs_f, rs_ref) and same(g_f, g_ref)
        assert _rel_l2(gf_f.cpu().numpy(), o_gf) <= 1e-3
    assert same(gf_f, gf_ref)
    again = ops.render_backward(torch.from_numpy(grad_out).to(DEV), idx, qv, None, scaler, d["points"], d["radii"], vis,
                                d["first"], d["num"], radii_s, clip)
    assert torch.equal(again[0], gf_f) and torch.equal(again[1], g_f)  # deterministic
    _, g_only = ops.render_backward(torch.from_numpy(grad_out).to(DEV), idx, qv, wsum, scaler, d["points"], d["radii"],
                                    vis, d["first"], d["num"], radii_s, clip, with_features=False)
    assert torch.equal(g_only, g_f)


@pytest.mark.parametrize("name", CASES)
def test_backward_pieces_vs_oracle(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    S, K, radii_s = int(z["S"]), int(z["K"]), float(z["radii_s"])
    P = z["points"].shape[0]
    d = _dev(z)
    idx = torch.from_numpy(z["ref_idx"]).to(DEV)
    vis = oracle.visibility(z["ref_idx"], P)
    o_rs = oracle.backward_radius(z["radii"], vis, z["first_idx"], z["num_pts"], radii_s)
    rs = ops.backward_radius(d["radii"], torch.from_numpy(vis).to(DEV), d["first"], d["num"], radii_s)
    assert np.array_equal(rs.cpu().numpy(), o_rs)
    g = ops.occ_backward(d["points"], d["radii"], torch.from_numpy(vis).to(DEV), rs,
                         torch.from_numpy(z["grad_occ"]).to(DEV), d["first"], d["num"])
    o_g = oracle.occ_backward_fast(z["points"], z["radii"], vis, o_rs, z["grad_occ"], z["first_idx"], z["num_pts"])
    assert _rel_l2(g[:, :2].cpu().numpy(), o_g) <= 1e-5
    assert (g[:, 2] == 0).all()
    # deterministic: same bits twice
    g2 = ops.occ_backward(d["points"], d["radii"], torch.from_numpy(vis).to(DEV), rs,
                          torch.from_numpy(z["grad_occ"]).to(DEV), d["first"], d["num"])
    assert torch.equal(g, g2)
    # zbuf backward vs the REFERENCE output (pinned)
    gz = torch.zeros(P, 1, device=DEV)
    ops._backward_zbuf(idx, torch.from_numpy(z["grad_zbuf"]).to(DEV), gz)
    assert np.allclose(gz[:, 0].cpu().numpy(), z["ref_grad_z"], rtol=1e-5, atol=1e-5)
    # row bands of the occupancy backward sum to the full result
    halves = [ops.occ_backward(d["points"], d["radii"], torch.from_numpy(vis).to(DEV), rs,
                               torch.from_numpy(z["grad_occ"][:, a:b]).to(DEV), d["first"], d["num"],
                               image_size=S, rows=(a, b)) for a, b in ((0, S // 3), (S // 3, S))]
    assert _rel_l2((halves[0] + halves[1]).cpu().numpy(), g.cpu().numpy()) <= 1e-5


FAST_CASES = [("ref_random64x2", 5.0), ("ref_random64x2", 1.0), ("ref_teapot256", 5.0), ("ref_teapot256", 1.0),
              ("ref_ties32", 5.0), ("ref_ties32", 1.0), ("ragged3", 5.0), ("ragged3", 2.0)]


@pytest.mark.parametrize("name,radii_s", FAST_CASES)
def test_backward_vs_executed_reference_cuda_kernel_golden(golden_dir, name, radii_s):
    """HIP backward (dss_splat_backward and the fused dss_render_backward) against tests/golden/ref_fast_backward.npz =
    the reference's own EllipticalRasterizer.backward around its fast CUDA kernel (rasterize_points_backward.cu:30-212,
    host-compiled and EXECUTED, tests/golden/make_golden_fast_backward.py).  rel-L2 <= 1e-3 (observed ~1e-7); points of
    the last grid cell of clouds n >= 1 are excluded: the reference skips them (:124-126), we do not."""
    g = np.load(os.path.join(golden_dir, "ref_fast_backward.npz"))
    if name == "ragged3":
        sc = {k: g["ragged3_" + k] for k in ("points", "ellipse", "cutoff", "radii", "first_idx", "num_pts")}
        idx_np, gocc, gz = g["ragged3_idx"], g["ragged3_grad_occ"], g["ragged3_grad_zbuf"]
    else:
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        sc = {k: z[k] for k in ("points", "ellipse", "cutoff", "radii", "first_idx", "num_pts")}
        idx_np, gocc, gz = z["ref_idx"], z["grad_occ"], z["grad_zbuf"]
    ref, lastcell = g["%s_s%g_grad" % (name, radii_s)], g["%s_s%g_lastcell" % (name, radii_s)]
    keep = ~lastcell
    P = sc["points"].shape[0]
    d = _dev(sc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    idx = t(idx_np)
    vis = t(oracle.visibility(idx_np, P))
    got = ops.splat_backward(d["points"], d["radii"], vis, idx, t(gocc), t(gz), d["first"], d["num"], radii_s).cpu().numpy()
    assert _rel_l2(got[keep, :2], ref[keep, :2]) <= 1e-3 and np.allclose(got[:, 2], ref[:, 2], rtol=1e-5, atol=1e-5)
    assert np.all(got[~vis.cpu().numpy().astype(bool)] == 0)
    if idx_np.shape[-1] <= 8:
        # fused kernel: occupancy gradient read in place from the alpha channel of an image gradient
        N, S, _, K = idx_np.shape
        go = np.zeros((N, S, S, 4), np.float32)
        go[..., 3] = gocc
        qv = t(np.where(idx_np >= 0, 0.5, -1.0).astype(np.float32))
        scaler = torch.ones(P, device=DEV)
        _, gp = ops.render_backward(t(go), idx, qv, None, scaler, d["points"], d["radii"], vis, d["first"], d["num"],
                                    radii_s, -1.0)
        assert _rel_l2(gp.cpu().numpy()[keep, :2], ref[keep, :2]) <= 1e-3
    if lastcell.any():
        assert np.all(ref[lastcell, :2] == 0)   # the documented reference behaviour


@pytest.mark.parametrize("name", ["ref_random64x2", "ref_teapot256", "ref_ties32"])
def test_blend_vs_executed_reference_renderer_golden(golden_dir, name):
    """HIP blend (dss_blend_forward / dss_blend_backward{,_scatter}, the fused dss_render_backward and the renderer
    class with and without compositor) against tests/golden/ref_blend.npz = the reference's own
    SurfaceSplattingRenderer.forward + gather_with_neg_idx + its weighted-sum CUDA kernels (weighted_sum.cu:38-134)
    host-compiled and EXECUTED (tests/golden/make_golden_blend.py).  RGB <= 1e-4, feature-gradient rel-L2 <= 1e-3."""
    from dss_amd.rasterizer import PointFragments
    from dss_amd.renderer import NormWeightedCompositor, SurfaceSplattingRenderer
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    g = np.load(os.path.join(golden_dir, "ref_blend.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    idx, qv, occ, scaler, feat = t(z["ref_idx"]), t(z["ref_qvalue"]), t(z["ref_occ"]), t(z["scaler"]), t(z["colors"])
    P = feat.shape[0]
    go = t(g[name + "_grad_out"])
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, feat, return_wsum=True)
    assert np.abs(img.cpu().numpy() - g[name + "_norm_image"]).max() <= 1e-4
    ref_gf = g[name + "_norm_grad_features"]
    d = _dev(z)
    vis = t(oracle.visibility(z["ref_idx"], P))
    geom = (d["points"], d["radii"], vis, d["first"], d["num"])
    for kw in (dict(), dict(geometry=geom), dict(geometry=geom, wsum=wsum)):
        gf, gocc = ops.blend_backward(go, idx, qv, scaler, P, **kw)
        assert _rel_l2(gf.cpu().numpy(), ref_gf) <= 1e-3, list(kw)
        assert torch.equal(gocc, go[..., 3])
    if z["ref_idx"].shape[-1] <= 8:
        gf, _ = ops.render_backward(go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"], 5.0, -1.0)
        assert _rel_l2(gf.cpu().numpy(), ref_gf) <= 1e-3

    # the renderer class, handed the fragments like renderer.py:44-50 (`fragments=`), both compositor settings
    class _Raster:
        cameras = None

    class _Cloud:
        def __init__(self, f):
            self.f = f

        def isempty(self):
            return False

        def features_packed(self):
            return self.f

    for comp, tag in ((NormWeightedCompositor(), "norm"), (None, "ws")):
        f = feat.clone().requires_grad_(True)
        frag = PointFragments(idx=idx, zbuf=t(z["ref_zbuf"]), qvalue=qv, scaler=scaler, occupancy=occ)
        out = SurfaceSplattingRenderer(_Raster(), comp)(_Cloud(f), fragments=frag)
        ref_img = g["%s_%s_image" % (name, tag)]
        assert np.abs(out.detach().cpu().numpy() - ref_img).max() <= 1e-4 * max(1.0, float(np.abs(ref_img).max()))
        (out * go).sum().backward()
        assert _rel_l2(f.grad.cpu().numpy(), g["%s_%s_grad_features" % (name, tag)]) <= 1e-3, tag
        # the reference-style per-fragment scaler (N,H,W,K) (rasterizer.py:631-633) is accepted as well
        frag4 = PointFragments(idx=idx, zbuf=t(z["ref_zbuf"]), qvalue=qv, scaler=t(g[name + "_frag_scaler"]), occupancy=occ)
        out4 = SurfaceSplattingRenderer(_Raster(), comp)(_Cloud(feat), fragments=frag4)
        assert np.abs(out4.cpu().numpy() - ref_img).max() <= 1e-4 * max(1.0, float(np.abs(ref_img).max()))


def test_dss_c_same_name_mirrors(golden_dir):
    """The four `DSS._C` exports beyond splat_points / _splat_points_naive / _backward_zbuf (ext.cpp:10-12, 14), under
    their reference names and argument order in dss_amd.ops: coarse + fine == splat_points bit for bit; the fast CUDA
    backward on sorted visible points == the executed reference kernel's golden; the box-supported slow backward ==
    the executed RasterizePointsOccBackwardCudaKernel golden."""
    g = np.load(os.path.join(golden_dir, "ref_fast_backward.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    for name in ("ref_random64x2", "ref_teapot256"):
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        S, K, thr = int(z["S"]), int(z["K"]), float(z["thr"])
        d = _dev(z)
        bins = ops._rasterize_coarse(d["points"], d["radii"], d["first"], d["num"], S, 16, 10000)
        out = ops._rasterize_fine(d["points"], d["ellipse"], d["cutoff"], d["radii"], bins, thr, S, 16, K)
        for a, k in zip(out, ("ref_idx", "ref_zbuf", "ref_qvalue", "ref_occ")):
            assert np.array_equal(a.cpu().numpy(), z[k]), k
        with pytest.raises(RuntimeError):
            ops._rasterize_fine(d["points"], d["ellipse"], d["cutoff"], d["radii"], torch.zeros(8, device=DEV), thr, S, 16, K)
        # the lists' metadata follows the STORAGE (views, detach(), autograd saves keep working); a copy is refused, and so
        # are lists built from other points
        out_v = ops._rasterize_fine(d["points"], d["ellipse"], d["cutoff"], d["radii"], bins.detach().view(-1), thr, S, 16, K)
        assert torch.equal(out_v[0], out[0])
        with pytest.raises(RuntimeError, match="view of it"):
            ops._rasterize_fine(d["points"], d["ellipse"], d["cutoff"], d["radii"], bins.clone(), thr, S, 16, K)
        with pytest.raises(RuntimeError, match="other points"):
            ops._rasterize_fine(d["points"].clone(), d["ellipse"], d["cutoff"], d["radii"], bins, thr, S, 16, K)
        # non-contiguous inputs (normalised into a temporary inside each call) are recognised as the same
        # tensors, and a view of the lists WITH a storage offset finds its metadata (keyed by the storage)
        pts_nc = torch.cat([d["points"], d["points"]], dim=1)[:, :3]
        rad_nc = torch.cat([d["radii"], d["radii"]], dim=1)[:, :2]
        assert not pts_nc.is_contiguous() and not rad_nc.is_contiguous()
        bins_nc = ops._rasterize_coarse(pts_nc, rad_nc, d["first"], d["num"], S, 16, 10000)
        out_nc = ops._rasterize_fine(pts_nc, d["ellipse"], d["cutoff"], rad_nc, bins_nc[16:], thr, S, 16, K)
        assert torch.equal(out_nc[0], out[0]) and torch.equal(out_nc[2], out[2])
        # _splat_points_occ_backward (CUDA form): every point, box support radii * radii_s
        got = ops._splat_points_occ_backward(d["points"], d["radii"], t(z["grad_occ"]), d["first"], d["num"],
                                             float(z["radii_s"]), thr)
        assert tuple(got.shape) == (z["points"].shape[0], 2)
        assert _rel_l2(got.cpu().numpy(), g[name + "_slowcuda_grad"]) <= 1e-5
        # _splat_points_occ_fast_cuda_backward called like rasterizer.py:951-952: visible points only, any order
        P = z["points"].shape[0]
        vis = oracle.visibility(z["ref_idx"], P)
        ids = np.nonzero(vis)[0]
        rng = np.random.default_rng(3)
        first_v = np.concatenate([[0], np.cumsum([vis[f:f + n].sum() for f, n in zip(z["first_idx"], z["num_pts"])])[:-1]])
        num_v = np.array([vis[f:f + n].sum() for f, n in zip(z["first_idx"], z["num_pts"])], np.int64)
        order = np.concatenate([f + rng.permutation(n) for f, n in zip(first_v, num_v)]).astype(np.int64)  # "sorted" order
        sel = ids[order]
        for radii_s in (5.0, 1.0):
            rs = oracle.backward_radius(z["radii"], vis, z["first_idx"], z["num_pts"], radii_s)
            gs = ops._splat_points_occ_fast_cuda_backward(t(z["points"][sel]), t(z["radii"][sel]), t(rs), t(z["grad_occ"]),
                                                          t(num_v), t(first_v.astype(np.int64)), None, None)
            ref = g["%s_s%g_grad" % (name, radii_s)]
            keep = ~g["%s_s%g_lastcell" % (name, radii_s)][sel]
            assert _rel_l2(gs.cpu().numpy()[keep], ref[sel][keep, :2]) <= 1e-5


def test_integration_stub_file_matches_the_python_mirror(golden_dir):
    """`integration/DSS_C.py` -- the file INTEGRATION.md section 3 tells a maintainer to drop in as `DSS/_C.py`: plain
    ctypes on the C ABI, no `dss_amd` import -- executed: all seven `DSS._C` names, with the reference's argument order,
    give what `dss_amd.ops` gives (which the tests above pin to the reference's goldens)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("DSS_C_stub", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(
        __file__))), "integration", "DSS_C.py"))
    C = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(C)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    z = np.load(os.path.join(golden_dir, "ref_random64x2.npz"))
    S, K, thr = int(z["S"]), int(z["K"]), float(z["thr"])
    d = _dev(z)
    want = ops.splat_points(d["points"], d["ellipse"], d["cutoff"], d["radii"], d["first"], d["num"], thr, S, K, None, None)
    for got in (C.splat_points(d["points"], d["ellipse"], d["cutoff"], d["radii"], d["first"], d["num"], thr, S, K, None, None),
                C._splat_points_naive(d["points"], d["ellipse"], d["cutoff"], d["radii"], d["first"], d["num"], thr, S, K),
                C._rasterize_fine(d["points"], d["ellipse"], d["cutoff"], d["radii"],
                                  C._rasterize_coarse(d["points"], d["radii"], d["first"], d["num"], S, 16, 10000), thr, S, 16, K)):
        for a, b, k in zip(got, want, ("idx", "zbuf", "qvalue", "occupancy")):
            assert torch.equal(a, b), k
            assert np.array_equal(a.cpu().numpy(), z["ref_" + ("occ" if k == "occupancy" else k)]), k
    go = t(z["grad_occ"])
    assert torch.equal(C._splat_points_occ_backward(d["points"], d["radii"], go, d["first"], d["num"], float(z["radii_s"]), thr),
                       ops._splat_points_occ_backward(d["points"], d["radii"], go, d["first"], d["num"], float(z["radii_s"]), thr))
    P = z["points"].shape[0]
    vis = oracle.visibility(z["ref_idx"], P)
    sel = np.nonzero(vis)[0]
    num_v = np.array([vis[f:f + n].sum() for f, n in zip(z["first_idx"], z["num_pts"])], np.int64)
    first_v = np.cumsum(num_v) - num_v
    rs = t(oracle.backward_radius(z["radii"], vis, z["first_idx"], z["num_pts"], 5.0))
    a = (t(z["points"][sel]), t(z["radii"][sel]), rs, go, t(num_v), t(first_v), None, None)
    assert torch.equal(C._splat_points_occ_fast_cuda_backward(*a), ops._splat_points_occ_fast_cuda_backward(*a))
    gz = torch.randn((int(z["num_pts"].shape[0]), S, S, K), generator=torch.Generator().manual_seed(2)).to(DEV)
    za, zb = torch.zeros((P, 1), device=DEV), torch.zeros((P, 1), device=DEV)
    C._backward_zbuf(want[0], gz, za)
    ops._backward_zbuf(want[0], gz, zb)
    assert torch.allclose(za, zb, rtol=1e-5, atol=1e-6) and float(za.abs().sum()) > 0


def test_point_on_pixel_centre_contributes_zero():
    """point-one KAT: a point exactly on a pixel centre (reference: 0/0 = NaN, documented divergence)."""
    S = 8
    ndc = np.float32(-1) + np.float32(2 * 3 + 1) / np.float32(S)
    pts = torch.tensor([[ndc, ndc, 1.0]], device=DEV)
    radii = torch.full((1, 2), 0.3, device=DEV)
    vis = torch.ones(1, dtype=torch.bool, device=DEV)
    first = torch.zeros(1, dtype=torch.int64, device=DEV)
    num = torch.ones(1, dtype=torch.int64, device=DEV)
    rs = torch.tensor([0.01], device=DEV)  # only the centre pixel is in range
    g = ops.occ_backward(pts, radii, vis, rs, torch.ones(1, S, S, device=DEV), first, num)
    assert torch.equal(g, torch.zeros(1, 3, device=DEV))


def test_clip_grad_matches_torch_hook():
    g = torch.randn(1000, 3, device=DEV) * 0.1
    g[0] = 0
    want = torch.nn.functional.normalize(g, dim=-1) * g.norm(dim=-1, keepdim=True).clamp(0, 0.05)
    got = ops.clip_grad_(g.clone(), 0.05)
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("C", [1, 3, 4, 8])
def test_blend_generic_channels(C):
    sc = scenes.random_splats(800, 40, 2, seed=C)
    idx, zbuf, qv, occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"],
                                              sc["num_pts"], 40, 5, 0.5)
    rng = np.random.default_rng(C)
    feat = rng.uniform(0, 1, (sc["points"].shape[0], C)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    img = ops.blend_forward(t(idx), t(qv), t(occ), t(sc["scaler"]), t(feat))
    assert np.abs(img.cpu().numpy() - oracle.blend_forward(idx, qv, occ, sc["scaler"], feat)).max() <= 1e-5
    go = rng.standard_normal((2, 40, 40, C + 1)).astype(np.float32)
    o_gf, o_gocc = oracle.blend_backward(go, idx, qv, sc["scaler"], feat.shape[0])
    vis = t(oracle.visibility(idx, feat.shape[0]))
    geom = (t(sc["points"]), t(sc["radii"]), vis, t(sc["first_idx"]), t(sc["num_pts"]))
    for kw in (dict(), dict(geometry=geom)):
        gf, gocc = ops.blend_backward(t(go), t(idx), t(qv), t(sc["scaler"]), feat.shape[0], **kw)
        assert _rel_l2(gf.cpu().numpy(), o_gf) <= 1e-5 and np.array_equal(gocc.cpu().numpy(), o_gocc)


def test_large_synthetic_properties():
    """BASELINE config 4 scale (1M points, 1024^2, here 2 of the 8 ring cameras): properties that do
    not need the oracle -- sortedness, depth-merge bound, occupancy == first slot filled, every
    fragment passes the hit test, row bands == full image, determinism."""
    P, S, K, thr = 1_000_000, 1024, 5, 0.05
    pts, nrm, col = scenes.synthetic_cloud(P, seed=0)
    M, V, _ = scenes.camera_matrices(2.0, 20.0, [0.0, 45.0])
    sc = scenes.setup_scene(pts, nrm, M, V, S, h=2e-5 * 4, colors=col)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, S, K, thr, return_visible=True)
    idx2 = _fwd(d, S, K, thr)[0]
    assert torch.equal(idx, idx2)
    valid = idx >= 0
    assert torch.equal(occ > 0, valid[..., 0])
    assert (valid[..., 1:] <= valid[..., :-1]).all()  # packed front to back
    zz = torch.where(valid, zbuf, torch.full_like(zbuf, float("inf")))
    assert (zz[..., 1:] >= zz[..., :-1]).all()
    assert ((zbuf - zbuf[..., :1])[valid] <= thr).all()
    # hit test of every fragment, recomputed with torch in fp32
    n, r, c, k = valid.nonzero(as_tuple=True)
    p = idx[valid].long()
    ndc = lambda i: -1 + (2 * (S - 1 - i).float() + 1.0) / S
    dx, dy = ndc(c) - d["points"][p, 0], ndc(r) - d["points"][p, 1]
    assert (dx.abs() <= d["radii"][p, 0]).all() and (dy.abs() <= d["radii"][p, 1]).all()
    first, num = d["first"], d["num"]
    assert ((p >= first[n]) & (p < first[n] + num[n])).all()
    assert torch.equal(vis, torch.zeros_like(vis).index_fill_(0, p, True))
    band = _fwd(d, S, K, thr, rows=(256, 640))
    assert torch.equal(band[0], idx[:, 256:640]) and torch.equal(band[3], occ[:, 256:640])
    assert occ.mean().item() > 0.2


def test_bench_two_rank_row_partition_matches_single_rank(tmp_path):
    """The N>1 path of bench.py (row bands + all-gather + visibility/grad reductions) on ONE GPU:
    two ranks share cuda:0 over gloo; the gathered image and the reduced gradients must equal the
    single-rank step (image bit-exact, gradients to fp32 reduction order)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(str(tmp_path), "two_rank.py")
    open(script, "w").write('''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("gloo")
# the reference: the same causal step (render -> image loss -> backward) on one rank, no collective
ref = bench.Workload(dev, world, bench.RowPartition(bench.S, 1, 0), multi="local")
img1, gw1, gc1 = [t.clone() for t in ref.step()]
rel = lambda a, b: float((a - b).norm() / b.norm())
# gradient exchange: "owner" (round 5: whole position gradients on the owner of a point's centre row, clip + projection in
# front of ONE all-reduce of the world-space sums) and "bucket" (partial sums of every pair reduced first)
for cyclic, grad in ((False, "owner"), (True, "owner"), (False, "bucket"), (True, "bucket")):
    os.environ["BENCH_GRADIENT"] = grad
    wl = bench.Workload(dev, world, bench.RowPartition(bench.S, world, rank, cyclic=cyclic))
    assert wl.owner == (grad == "owner")
    img, gw, gc = wl.step()
    torch.cuda.synchronize()
    assert torch.equal(img, img1), "gathered image differs from the single-rank render (cyclic=%%s)" %% cyclic
    assert rel(gw, gw1) < 1e-5 and rel(gc, gc1) < 1e-5, (cyclic, rel(gw, gw1), rel(gc, gc1))
    # the same step with its compute segments replayed as graphs around the collectives
    wl.capture_segments()
    for _ in range(2):
        img, gw, gc = wl.step_segments()
    torch.cuda.synchronize()
    assert torch.equal(img, img1), "graph-segment step: image differs (cyclic=%%s)" %% cyclic
    assert rel(gw, gw1) < 1e-5 and rel(gc, gc1) < 1e-5, ("segments", cyclic, rel(gw, gw1), rel(gc, gc1))
    # round 5: the folded exchange (two collectives: the visibility flags ride in the image all-gather), eager + segments
    wl.set_exchange(True)
    assert wl.fx.fold and wl.fx.vrows >= 1
    img, gw, gc = wl.step()
    torch.cuda.synchronize()
    assert torch.equal(img, img1), "folded exchange: image differs (cyclic=%%s)" %% cyclic
    assert rel(gw, gw1) < 1e-5 and rel(gc, gc1) < 1e-5, ("fold", cyclic, rel(gw, gw1), rel(gc, gc1))
    wl.capture_segments()
    for _ in range(2):
        img, gw, gc = wl.step_segments()
    torch.cuda.synchronize()
    assert torch.equal(img, img1), "folded exchange, graph segments: image differs (cyclic=%%s)" %% cyclic
    assert rel(gw, gw1) < 1e-5 and rel(gc, gc1) < 1e-5, ("fold segments", cyclic, rel(gw, gw1), rel(gc, gc1))
open(os.path.join(%r, "ok%%d" %% rank), "w").write("%%g %%g" %% (rel(gw, gw1), rel(gc, gc1)))
dist.destroy_process_group()
''' % (root, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29711", script],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-12000:]
    assert os.path.exists(os.path.join(str(tmp_path), "ok0")) and os.path.exists(os.path.join(str(tmp_path), "ok1"))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start its two ranks itself (the driver calls the
    N>1 runs exactly like the N=1 run) and print ONE JSON line that records the world size torch.distributed saw.
    One GPU here, so the two ranks share cuda:0 over gloo (BENCH_DIST_BACKEND); on a multi-GPU box the default is RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_DIST_BACKEND"] = "gloo"
    env["BENCH_EXCHANGE"] = "overlap"   # (the three-collective form: "auto" would time both forms and keep the faster)
    env["BENCH_GRADIENT"] = "owner"     # (the form with the alpha-plane exchange; the default "auto" picks the bucket form at this size)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-12000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    d = rec["config"]["dist"]
    assert rec["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo"
    # diagnosable multi-GPU line: the communicator set-up that was used and where the step's time went, per rank
    assert d["overlap"] is True and d["degraded"] is None and d["visible_devices"] >= 1 and "cyclic" in d["partition"]
    # visibility, loss sums, alpha-gradient plane (owner form, two ranks), gradient sums, image
    assert d["exchange"]["form"] == "overlap" and d["collectives_per_step"] == 5 and d["causal"] is True
    assert rec["config"]["launch"] == "graph_segments" and d["segment_capture"] == "ok"
    t = d["timing_us"]
    for k in ("forward_compute", "loss_sums_compute", "loss_gradient_compute", "backward_compute", "compute_us",
              "wait_visibility_allreduce", "wait_loss_allreduce", "wait_alpha_allgather", "wait_gradient_allreduce",
              "wait_image_allgather"):
        assert t[k]["min"] <= t[k]["mean"] <= t[k]["max"] and t[k]["max"] > 0, (k, t[k])
    assert rec["config"]["cameras"] == 2 and rec["value"] > 0
    # a launcher whose world size disagrees with --gpus is an error, not a silent mismatch
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True,
                         timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stdout + bad.stderr)


def test_render_backward_multi_cloud_matches_unfused():
    sc = scenes.random_splats(3000, 96, 3, seed=11)
    sc["first_idx"] = np.array([0, 3000, 6500], np.int64)     # ragged clouds + a gap of unowned points
    sc["num_pts"] = np.array([2900, 3500, 2500], np.int64)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, 96, 5, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    feat = torch.from_numpy(sc["colors"]).to(DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, feat, return_wsum=True)
    go = torch.randn_like(img)
    P = sc["points"].shape[0]
    geom = (d["points"], d["radii"], vis, d["first"], d["num"])
    gf_ref, gocc = ops.blend_backward(go, idx, qv, scaler, P, geometry=geom, wsum=wsum)
    g_ref = ops.splat_backward(d["points"], d["radii"], vis, idx, gocc, None, d["first"], d["num"], 4.0, 0.05)
    gf, g = ops.render_backward(go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"], 4.0, 0.05)
    assert _rel_l2(gf.cpu().numpy(), gf_ref.cpu().numpy()) <= 2e-6 and _rel_l2(g.cpu().numpy(), g_ref.cpu().numpy()) <= 2e-6
    o_g, _, _ = oracle.splat_backward(sc["points"], sc["radii"], idx.cpu().numpy(), go[..., 3].cpu().numpy(), None,
                                      sc["first_idx"], sc["num_pts"], 4.0, 0.05)
    assert _rel_l2(g.cpu().numpy(), o_g) <= 1e-4


@pytest.mark.parametrize("radii_s", [0.4, 1.5, 4.0, 12.0])
def test_render_backward_search_radius_sweep(radii_s):
    """radii_backward_scaler decays during training (scheduler.py:36-48): tiny to very large windows."""
    sc = scenes.random_splats(4000, 128, 2, seed=21, rmin=1.0, rmax=2.5)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, 128, 5, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    feat = torch.from_numpy(sc["colors"]).to(DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, feat, return_wsum=True)
    go = torch.randn_like(img)
    P = sc["points"].shape[0]
    gf, g, rs = ops.render_backward(go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"],
                                    radii_s, -1.0, return_rs=True)
    o_g, o_vis, o_rs = oracle.splat_backward(sc["points"], sc["radii"], idx.cpu().numpy(), go[..., 3].cpu().numpy(),
                                             None, sc["first_idx"], sc["num_pts"], radii_s, -1.0)
    o_gf, _ = oracle.blend_backward(go.cpu().numpy(), idx.cpu().numpy(), qv.cpu().numpy(), sc["scaler"], P)
    assert np.array_equal(rs.cpu().numpy(), o_rs)
    assert _rel_l2(g.cpu().numpy(), o_g) <= 1e-4 and _rel_l2(gf.cpu().numpy(), o_gf) <= 1e-4


def test_empty_row_band_is_a_no_op():
    """RowPartition can hand a rank an empty band (e.g. S=10 over 8 ranks gives row0 == row1): every op short-circuits
    instead of rejecting row0 >= row1 -- no fragments, nothing visible, zero partial gradients."""
    sc = scenes.random_splats(500, 40, 2, seed=5)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, 40, 5, 0.3, rows=(16, 16), return_visible=True)
    assert tuple(idx.shape) == (2, 0, 40, 5) and tuple(occ.shape) == (2, 0, 40) and not bool(vis.any())
    full = _fwd(d, 40, 5, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    go = torch.zeros((2, 0, 40, 4), device=DEV)
    gf, gp, rs = ops.render_backward(go, idx, qv, None, scaler, d["points"], d["radii"], full[4], d["first"], d["num"], 4.0,
                                     -1.0, image_size=40, rows=(16, 16), return_rs=True)
    assert not bool(gf.any()) and not bool(gp.any())
    assert torch.equal(rs, ops.backward_radius(d["radii"], full[4], d["first"], d["num"], 4.0))
    g = ops.occ_backward(d["points"], d["radii"], full[4], rs, torch.zeros((2, 0, 40), device=DEV), d["first"], d["num"],
                         image_size=40, rows=(16, 16))
    assert not bool(g.any())


@pytest.mark.parametrize("S,bounds", [(96, (0, 40, 96)), (96, (0, 8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96)),
                                      (50, (0, 5, 38, 50)), (128, (0, 127, 128))])
def test_render_backward_row_bands_sum_to_full(S, bounds):
    """Multi-GPU contract of the fused backward: band partial sums (global visibility, no clip) add up to
    the full-image result -- equal and odd bands, image sizes that are not powers of two or multiples of the
    tile, a one-row band; the band filter drops the points that cannot reach a band."""
    sc = scenes.random_splats(3000, S, 2, seed=31)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, S, 5, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, torch.from_numpy(sc["colors"]).to(DEV), return_wsum=True)
    go = torch.randn_like(img)
    gf, g = ops.render_backward(go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"], 4.0, -1.0)
    sum_g, sum_gf = torch.zeros_like(g), torch.zeros_like(gf)
    for a, b in zip(bounds[:-1], bounds[1:]):
        pf, pg = ops.render_backward(go[:, a:b].contiguous(), idx[:, a:b].contiguous(), qv[:, a:b].contiguous(),
                                     wsum[:, a:b].contiguous(), scaler, d["points"], d["radii"], vis, d["first"],
                                     d["num"], 4.0, -1.0, image_size=S, rows=(a, b))
        sum_g += pg
        sum_gf += pf
    assert _rel_l2(sum_g.cpu().numpy(), g.cpu().numpy()) <= 1e-5
    assert _rel_l2(sum_gf.cpu().numpy(), gf.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("S,G,P", [(96, 2, 3000), (128, 4, 6000), (37, 2, 1500), (100, 4, 3000), (512, 8, 40000),
                                   (256, 8, 300000)])
def test_tile_row_cyclic_bands_reproduce_the_full_render_and_its_gradients(S, G, P):
    """Multi-GPU partition of VERDICT r2 item 3a: rank g renders the 8-row tile rows g, g + G, ... (`rows=(8 g, S, G)` ->
    `row_cycle` of dss_render_forward / dss_render_backward).  The bands' fragments, image and weight sums are the rows of
    the full render bit for bit, the visibility flags add up to the full set, and the backward partial sums (global
    visibility, clip deferred) add up to the full gradient -- image sizes that are not multiples of 8 G included, every
    tasks-per-wavefront variant of the gather, both preparation paths (P <= 262,144 and above)."""
    from dss_amd.distributed import RowPartition
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    reps = max(1, -(-P // len(pts)))
    if reps > 1:
        pts, nrm = scenes.upsample_jitter(pts, nrm, reps, seed=2)
    pts, nrm = pts[:P], nrm[:P]
    Pc = len(pts)
    h = scenes.global_h(pts[:: max(1, Pc // 20000)]) * (20000.0 / Pc if Pc > 20000 else 1.0)
    N = 2
    M = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[0] for a in (30.0, 200.0)])
    V = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[1] for a in (30.0, 200.0)])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    first = torch.arange(N, device=DEV, dtype=torch.int64) * Pc
    num = torch.full((N,), Pc, dtype=torch.int64, device=DEV)
    feat = torch.rand((N * Pc, 3), device=DEV)
    a = (t(pts), t(nrm), torch.full((N,), float(h), device=DEV), t(M), t(V), torch.full((N,), 0.1, device=DEV),
         torch.full((N,), 100.0, device=DEV), first, num, feat)
    K = 5
    full = ops.render_forward(*a, S, K, 1.0, 0.05, 1.0, False, True)
    assert float(full["occupancy"].mean()) > 0.02
    go = torch.randn((N, S, S, 4), device=DEV)
    bw = lambda f, g, vis, rows=None: ops.render_backward(g, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"],
                                                          f["radii"], vis, first, num, 4.0, -1.0, image_size=S, rows=rows)
    gf_full, gp_full = bw(full, go, full["visible"])
    bands, vis_sum = [], torch.zeros_like(full["visible"])
    for g in range(G):
        part = RowPartition(S, G, g, cyclic=True)
        own = torch.tensor(part.row_indices(), device=DEV, dtype=torch.int64)
        f = ops.render_forward(*a, S, K, 1.0, 0.05, 1.0, False, True, rows=part.rows)
        assert f["idx"].shape[1] == len(own) == part.n_rows
        for k in ("idx", "zbuf", "qvalue", "occupancy", "image", "wsum"):
            assert torch.equal(f[k], full[k][:, own]), (k, g)
        vis_sum |= f["visible"]
        bands.append((part, own, f))
    assert torch.equal(vis_sum, full["visible"])
    for tpw in (0, 1, 2, 4):
        _lib.set_option(_lib.OPT_BACKWARD_TPW, tpw)
        try:
            sum_f, sum_p = torch.zeros_like(gf_full), torch.zeros_like(gp_full)
            for part, own, f in bands:
                if len(own) == 0:
                    continue
                pf, pp = bw(f, go[:, own].contiguous(), full["visible"], rows=part.rows)
                sum_f += pf
                sum_p += pp
        finally:
            _lib.set_option(_lib.OPT_BACKWARD_TPW, 0)
        assert _rel_l2(sum_p.cpu().numpy(), gp_full.cpu().numpy()) <= 1e-5, tpw
        assert _rel_l2(sum_f.cpu().numpy(), gf_full.cpu().numpy()) <= 1e-5, tpw


def test_tile_row_cyclic_backward_refuses_what_it_was_not_built_for():
    """The tile-row-cyclic gather is built for the training configuration (RGB features, 32-bit offsets): another channel
    count or forced 64-bit addressing is DSS_ERR_UNSUPPORTED with a message, not a wrong gradient."""
    sc = scenes.random_splats(2000, 64, 1, seed=3)
    d = _dev(sc)
    S, K, G = 64, 5, 2
    idx, zbuf, qv, occ, vis = _fwd(d, S, K, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    own = torch.tensor([r for r in range(S) if (r // 8) % G == 0], device=DEV)
    band = lambda t: t[:, own].contiguous()
    for C, addr64 in ((1, 0), (5, 0), (3, 1)):
        feat = torch.rand((sc["points"].shape[0], C), device=DEV)
        img, wsum = ops.blend_forward(idx, qv, occ, scaler, feat, return_wsum=True)
        go = torch.randn_like(img)
        _lib.set_option(_lib.OPT_BACKWARD_ADDR64, addr64)
        try:
            with pytest.raises(RuntimeError, match="tile-row-cyclic band needs C == 3"):
                ops.render_backward(band(go), band(idx), band(qv), band(wsum), scaler, d["points"], d["radii"], vis,
                                    d["first"], d["num"], 4.0, -1.0, image_size=S, rows=(0, S, G))
        finally:
            _lib.set_option(_lib.OPT_BACKWARD_ADDR64, 0)
    # ... and the supported shape of the same call goes through
    feat = torch.rand((sc["points"].shape[0], 3), device=DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, feat, return_wsum=True)
    gf, gp = ops.render_backward(band(torch.randn_like(img)), band(idx), band(qv), band(wsum), scaler, d["points"],
                                 d["radii"], vis, d["first"], d["num"], 4.0, -1.0, image_size=S, rows=(0, S, G))
    assert torch.isfinite(gp).all() and torch.isfinite(gf).all()


@pytest.mark.parametrize("P,S,N", [(4000, 128, 1), (3000, 96, 3), (300000, 256, 2)])
def test_render_backward_with_fused_projection_equals_project_backward(P, S, N):
    """`project=(world, M)`: the backward of the projection (dss_project_backward) evaluated in the gather's epilogue --
    same arithmetic, so the world-space gradients equal the two-launch path bit for bit (clip applied first); clouds that
    are not shared between cameras, every tasks-per-wavefront variant, both preparation paths."""
    pts, nrm = scenes.load_cloud("bunny")
    pts = scenes.normalize_unit_sphere(pts)
    reps = max(1, -(-P // len(pts)))
    if reps > 1:
        pts, nrm = scenes.upsample_jitter(pts, nrm, reps, seed=4)
    pts, nrm = pts[:P], nrm[:P]
    Pc = len(pts)
    h = scenes.global_h(pts[:: max(1, Pc // 20000)]) * (20000.0 / Pc if Pc > 20000 else 1.0)
    az = [30.0, 150.0, 260.0][:N]
    M = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[0] for a in az])
    V = np.concatenate([scenes.camera_matrices(2.0, 20.0, a)[1] for a in az])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    world = t(np.tile(pts, (N, 1)) + np.repeat(np.arange(N, dtype=np.float32)[:, None] * 0.01, Pc, 0))   # N distinct clouds
    normals = t(np.tile(nrm, (N, 1)))
    first = torch.arange(N, device=DEV, dtype=torch.int64) * Pc
    num = torch.full((N,), Pc, dtype=torch.int64, device=DEV)
    feat = torch.rand((N * Pc, 3), device=DEV)
    f = ops.render_forward(world, normals, torch.full((N,), float(h), device=DEV), t(M), t(V),
                           torch.full((N,), 0.1, device=DEV), torch.full((N,), 100.0, device=DEV), first, num, feat, S, 5, 1.0,
                           0.05, 1.0, False, False)
    go = torch.randn((N, S, S, 4), device=DEV)
    a = (go, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], f["visible"], first, num, 4.0, 0.05)
    for tpw in (0, 1, 2, 4):   # (the lane tiling, hence the summation order, differs between the variants: compare per variant)
        _lib.set_option(_lib.OPT_BACKWARD_TPW, tpw)
        try:
            gf, gs = ops.render_backward(*a)
            want = ops.project_backward(world, t(M), t(V), first, num, gs, f["valid"], False)
            gf2, gw = ops.render_backward(*a, project=(world, t(M)))
        finally:
            _lib.set_option(_lib.OPT_BACKWARD_TPW, 0)
        assert float(want.abs().max()) > 0
        assert torch.equal(gf2, gf) and torch.equal(gw, want), tpw
    with pytest.raises(RuntimeError, match="fused projection"):
        ops.render_backward(go[:, :S // 2].contiguous(), f["idx"][:, :S // 2].contiguous(), f["qvalue"][:, :S // 2].contiguous(),
                            f["wsum"][:, :S // 2].contiguous(), *a[4:], image_size=S, rows=(0, S // 2), project=(world, t(M)))


def test_render_backward_row_bands_sum_to_full_on_the_long_list_path():
    """The same contract for more than 262,144 points: multi-kernel median, screen-cell order of the visible list, XCD-wise
    dealing -- what a rank of the 8-GPU run of BASELINE configs[3] executes on its band."""
    S, bounds = 256, (0, 32, 100, 256)
    sc = scenes.random_splats(300000, S, 2, seed=33, rmin=0.6, rmax=2.0)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, S, 5, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, torch.from_numpy(sc["colors"]).to(DEV), return_wsum=True)
    go = torch.randn_like(img)
    gf, g = ops.render_backward(go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"], 4.0, -1.0)
    sum_g, sum_gf = torch.zeros_like(g), torch.zeros_like(gf)
    for a, b in zip(bounds[:-1], bounds[1:]):
        pf, pg = ops.render_backward(go[:, a:b].contiguous(), idx[:, a:b].contiguous(), qv[:, a:b].contiguous(),
                                     wsum[:, a:b].contiguous(), scaler, d["points"], d["radii"], vis, d["first"],
                                     d["num"], 4.0, -1.0, image_size=S, rows=(a, b))
        sum_g += pg
        sum_gf += pf
    assert _rel_l2(sum_g.cpu().numpy(), g.cpu().numpy()) <= 1e-5
    assert _rel_l2(sum_gf.cpu().numpy(), gf.cpu().numpy()) <= 1e-5
    want, _, _ = oracle.splat_backward(sc["points"], sc["radii"], idx.cpu().numpy(), go[..., 3].cpu().numpy(), None,
                                       sc["first_idx"], sc["num_pts"], 4.0, -1.0)
    assert _rel_l2(g.cpu().numpy()[:, :2], want[:, :2]) <= 1e-3


@pytest.mark.parametrize("P,S", [(3000, 96), (300000, 256)])
def test_render_backward_64_bit_addressing_variant_matches(P, S):
    """Gathered tensors of 4 GB and more take a kernel variant with 64-bit addresses (four tasks per wavefront); forced
    here on small inputs, it must reproduce the 32-bit-offset kernels (both preparation paths)."""
    sc = scenes.random_splats(P, S, 2, seed=8, rmin=0.6, rmax=2.0) if P > 10000 else scenes.random_splats(P, S, 2, seed=8)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, S, 5, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, torch.from_numpy(sc["colors"]).to(DEV), return_wsum=True)
    go = torch.randn_like(img)
    a = (go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"], 4.0, 0.05)
    gf, g = ops.render_backward(*a)
    _lib.set_option(_lib.OPT_BACKWARD_ADDR64, 1)
    try:
        gf64, g64 = ops.render_backward(*a)
    finally:
        _lib.set_option(_lib.OPT_BACKWARD_ADDR64, 0)
    assert _rel_l2(g64.cpu().numpy(), g.cpu().numpy()) <= 1e-6 and _rel_l2(gf64.cpu().numpy(), gf.cpu().numpy()) <= 1e-6


@pytest.mark.parametrize("P,S", [(3000, 96), (300000, 256)])
def test_render_backward_gather_stage_alone_reproduces_the_full_call(P, S):
    """`dss_render_backward_gather` (second stage only, on the workspace and zero-filled gradients of a preceding full
    call; both preparation paths: P <= 262144 and above) gives the full call's result bit for bit."""
    sc = scenes.random_splats(P, S, 2, seed=5, rmin=0.6, rmax=2.0) if P > 10000 else scenes.random_splats(P, S, 2, seed=5)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, S, 5, 0.3, return_visible=True)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, torch.from_numpy(sc["colors"]).to(DEV), return_wsum=True)
    go = torch.randn_like(img)
    a = (go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"], 4.0, 0.05)
    gf, g, rs = ops.render_backward(*a, return_rs=True)
    want_gf, want_g = gf.clone(), g.clone()
    g[vis.bool()] = 7.0          # the second stage rewrites every visible row and leaves the zero rows alone
    gf[vis.bool()] = 7.0
    gf2, g2 = ops.render_backward(*a, out=(gf, g), gather_only_rs=rs)
    assert torch.equal(g2, want_g) and torch.equal(gf2, want_gf)


@pytest.mark.parametrize("sizes,frac,dist", [
    ((0,), 0.5, "lognormal"),               # empty cloud
    ((1,), 1.0, "lognormal"),               # single point, two radii
    ((777, 0, 3001), 0.3, "lognormal"),     # ragged, one empty cloud
    ((32768,), 0.4, "lognormal"),           # exactly the register-resident capacity of the median workgroup
    ((32769, 5), 0.4, "equal"),             # one past it (second streaming chunk), all radii equal
    ((100000, 31072), 0.25, "lognormal"),   # 131072 points = 64 segments of 2048
    ((120000, 11000), 0.9, "lognormal"),    # > 32768 visible points in one cloud: keys re-streamed per pass
    ((131073,), 0.25, "wide"),              # one past it: segments of 4096 points
    ((200000, 62144), 0.3, "lognormal"),    # 262144 points = 64 segments of 4096: upper edge of the two-launch path
    ((262145,), 0.2, "wide"),               # one past it: multi-kernel radix select
    ((4096, 4096), 0.0, "lognormal"),       # nothing visible: rs = 0
])
def test_backward_radius_and_compaction_sizes(sizes, frac, dist):
    """Median radius (rasterizer.py:885-888) through BOTH device paths (two-launch compaction + register/LDS radix
    select for P <= 262144, multi-kernel select above) against the oracle, bit for bit, at their size boundaries; and
    the fused backward's compaction (every visible point gets a gradient row, every invisible one zeros)."""
    rng = np.random.default_rng(sum(sizes) + len(sizes))
    P = int(sum(sizes))
    if dist == "lognormal":
        radii = np.exp(rng.normal(-4.0, 0.7, size=(P, 2))).astype(np.float32)
    elif dist == "equal":
        radii = np.full((P, 2), 0.0123, np.float32)
    else:
        radii = (10.0 ** rng.uniform(-30, 30, size=(P, 2))).astype(np.float32)
    vis = (rng.random(P) < frac).astype(np.uint8)
    num = np.asarray(sizes, np.int64)
    first = np.concatenate([[0], np.cumsum(num)[:-1]]).astype(np.int64)
    o_rs = oracle.backward_radius(radii, vis, first, num, 2.5)
    t = lambda a: torch.from_numpy(a).to(DEV)
    if P > 0:
        rs = ops.backward_radius(t(radii), t(vis), t(first), t(num), 2.5)
        assert np.array_equal(rs.cpu().numpy(), o_rs), (rs.cpu().numpy(), o_rs)
    # fused backward on an empty image gradient: checks rs again plus the compaction bookkeeping
    S, K, N = 16, 2, len(sizes)
    if P == 0:
        return
    pts = np.zeros((P, 3), np.float32)
    pts[:, :2] = rng.uniform(-0.9, 0.9, size=(P, 2))
    pts[:, 2] = 1.0
    grad_out = np.zeros((N, S, S, 4), np.float32)
    grad_out[..., 3] = 1.0
    idx = torch.full((N, S, S, K), -1, dtype=torch.int32, device=DEV)
    qv = torch.zeros((N, S, S, K), device=DEV)
    wsum = torch.ones((N, S, S), device=DEV)
    scaler = torch.ones(P, device=DEV)
    small_r = np.minimum(radii, 0.05).astype(np.float32)
    o_rs2 = oracle.backward_radius(small_r, vis, first, num, 2.5)
    gf, g, rs2 = ops.render_backward(t(grad_out), idx, qv, wsum, scaler, t(pts), t(small_r), t(vis), t(first), t(num),
                                     2.5, 0.0, return_rs=True)
    assert np.array_equal(rs2.cpu().numpy(), o_rs2)
    o_g = oracle.occ_backward_fast(pts, small_r, vis, o_rs2, grad_out[..., 3].copy(), first, num)
    gn = g.cpu().numpy()
    assert (gn[vis == 0] == 0).all() and (gf.cpu().numpy()[vis == 0] == 0).all()
    assert _rel_l2(gn[:, :2], o_g) <= 1e-5


@pytest.mark.parametrize("K,C,rmax", [(1, 3, 2.5), (8, 1, 2.5), (12, 3, 2.5), (5, 5, 14.0), (3, 3, 40.0)])
def test_render_backward_fragment_depths_channels_and_splat_sizes(K, C, rmax):
    """Fused backward vs the oracle outside the benchmark shape: K = 1 / 8 (register path of the prefetched blend
    pixel) / 12 (generic blend path), C != 3 (runtime channel count), splats larger than one sweep of the
    wavefront (bounding boxes of up to 80 pixels: multi-step blend gather, multi-column occupancy window)."""
    sc = scenes.random_splats(1500, 128, 2, seed=31 + K, rmin=1.0, rmax=rmax)
    d = _dev(sc)
    idx, zbuf, qv, occ, vis = _fwd(d, 128, K, 0.3, return_visible=True)
    P = sc["points"].shape[0]
    rng = np.random.default_rng(K * 10 + C)
    feat_np = rng.random((P, C)).astype(np.float32)
    scaler = torch.from_numpy(sc["scaler"]).to(DEV)
    img, wsum = ops.blend_forward(idx, qv, occ, scaler, torch.from_numpy(feat_np).to(DEV), return_wsum=True)
    assert img.shape[-1] == C + 1
    go = torch.randn_like(img)
    gf, g, rs = ops.render_backward(go, idx, qv, wsum, scaler, d["points"], d["radii"], vis, d["first"], d["num"],
                                    3.0, 0.05, return_rs=True)
    o_g, o_vis, o_rs = oracle.splat_backward(sc["points"], sc["radii"], idx.cpu().numpy(), go[..., C].cpu().numpy(),
                                             None, sc["first_idx"], sc["num_pts"], 3.0, 0.05)
    o_gf, _ = oracle.blend_backward(go.cpu().numpy(), idx.cpu().numpy(), qv.cpu().numpy(), sc["scaler"], P)
    assert np.array_equal(rs.cpu().numpy(), o_rs)
    assert _rel_l2(g.cpu().numpy(), o_g) <= 1e-4 and _rel_l2(gf.cpu().numpy(), o_gf) <= 1e-4
    # and the unfused entry points give the same values (other summation order)
    geom = (d["points"], d["radii"], vis, d["first"], d["num"])
    gf_ref, gocc = ops.blend_backward(go, idx, qv, scaler, P, geometry=geom, wsum=wsum)
    g_ref = ops.splat_backward(d["points"], d["radii"], vis, idx, gocc, None, d["first"], d["num"], 3.0, 0.05)
    assert _rel_l2(gf.cpu().numpy(), gf_ref.cpu().numpy()) <= 2e-6 and _rel_l2(g.cpu().numpy(), g_ref.cpu().numpy()) <= 2e-6
