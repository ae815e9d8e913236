"""CPU, world_size 2 over gloo: the row-partition exchange steps of dss_amd.distributed.
The per-band compute is supplied by the oracle (the HIP kernels need a GPU); what is tested here is
that bands reassemble to the single-process image and that the backward exchanges (visibility
union, gradient partial sums) reproduce the single-process gradients."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, S, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import scenes
    from dss_amd.distributed import (ForwardExchange, GatherRows, OverlappedExchange, RowPartition, gather_rows_and_visibility,
                                     reduce_grads_, reduce_visibility_)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.random_splats(700, S, 2, seed=3)
        K, P = 4, sc["points"].shape[0]
        idx, zbuf, qv, occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"],
                                                  sc["first_idx"], sc["num_pts"], S, K, 0.5)
        full = oracle.blend_forward(idx, qv, occ, sc["scaler"], sc["colors"])
        part = RowPartition(S, world, rank)
        r0, r1 = part.rows
        # forward: each rank owns a band; all-gather must rebuild the single-process image exactly
        band = torch.from_numpy(full[:, r0:r1].copy()).requires_grad_(True)
        img = GatherRows.apply(band, part, None)
        assert torch.equal(img.detach(), torch.from_numpy(full)), "gathered image differs"
        g_full = torch.from_numpy(np.random.default_rng(5).standard_normal(full.shape).astype(np.float32))
        (img * g_full).sum().backward()
        assert torch.equal(band.grad, g_full[:, r0:r1])
        # backward: visibility union + gradient partial sums
        vis_band = torch.from_numpy(oracle.visibility(
            np.concatenate([np.full_like(idx[:, :r0], -1), idx[:, r0:r1], np.full_like(idx[:, r1:], -1)], 1), P)
            .astype(np.uint8))
        vis = reduce_visibility_(vis_band.clone(), part)
        assert np.array_equal(vis.numpy().astype(bool), oracle.visibility(idx, P))
        img2, vis2 = gather_rows_and_visibility(torch.from_numpy(full[:, r0:r1].copy()), vis_band.clone(), part)
        assert torch.equal(img2, torch.from_numpy(full)) and torch.equal(vis2, vis)
        fx = ForwardExchange(part, 2, full.shape[-1], P, "cpu")
        if fx.image is not None:  # zero-copy path: the producer writes into the send buffer
            fx.image.copy_(torch.from_numpy(full[:, r0:r1].copy()))
            fx.visible.copy_(vis_band)
            img3, vis3 = fx.exchange(fx.image)
        else:
            fx.visible.copy_(vis_band)
            img3, vis3 = fx.exchange(torch.from_numpy(full[:, r0:r1].copy()))
        assert torch.equal(img3, torch.from_numpy(full)) and torch.equal(vis3, vis)
        # overlapped variant (bench.py): image bands on a second group, asynchronous; (row, camera, col, ch)
        # send layout written through a strided (N, rows, S, ch) view; twice, to cover buffer reuse
        ox = OverlappedExchange(part, 2, full.shape[-1], P, "cpu")
        assert ox.image.shape == (2, r1 - r0, S, full.shape[-1]) and not ox.image.is_contiguous()
        for rep in range(2):
            ox.image.copy_(torch.from_numpy(full[:, r0:r1].copy()) * (rep + 1))
            ox.visible.copy_(vis_band)
            vis4 = ox.start()
            assert torch.equal(vis4, vis)
            img4 = ox.finish()
            assert img4.shape == full.shape and torch.equal(img4, torch.from_numpy(full) * (rep + 1))
        # load-balanced (unequal) bands: padded exchange, rows put in place by one gather
        from dss_amd.distributed import balanced_bounds
        wrow = torch.from_numpy(occ.sum(axis=(0, 2)))
        bnd = balanced_bounds(wrow, world, align=8, min_rows=8)
        assert bnd[0] == 0 and bnd[-1] == S and all(b % 8 == 0 for b in bnd[:-1])
        pb = RowPartition(S, world, rank, bounds=bnd)
        b0, b1 = pb.rows
        oxb = OverlappedExchange(pb, 2, full.shape[-1], P, "cpu")
        oxb.image.copy_(torch.from_numpy(full[:, b0:b1].copy()))
        oxb.visible.copy_(vis_band)
        oxb.start()
        assert torch.equal(oxb.finish(), torch.from_numpy(full))
        # tile-row-cyclic layout (rank g owns the 8-row tile rows g, g + G, ...): bands written through the strided send
        # view, interleaved back by the exchange; visibility union and gradient partial sums as for contiguous bands
        pc = RowPartition(S, world, rank, cyclic=True)
        own = np.array(pc.row_indices(), np.int64)
        assert pc.rows == (8 * rank, S, world) and len(own) == pc.n_rows and pc.band >= pc.n_rows
        full_t = torch.from_numpy(full)
        assert torch.equal(pc.slice(full_t), full_t[:, own])
        oxc = OverlappedExchange(pc, 2, full.shape[-1], P, "cpu")
        assert oxc.image.shape == (2, len(own), S, full.shape[-1])
        idx_c = np.full_like(idx, -1)
        idx_c[:, own] = idx[:, own]
        vis_c = torch.from_numpy(oracle.visibility(idx_c, P).astype(np.uint8))
        for rep in range(2):
            oxc.image.copy_(full_t[:, own] * (rep + 1))
            oxc.visible.copy_(vis_c)
            assert torch.equal(oxc.start(), vis)
            assert torch.equal(oxc.finish(), full_t * (rep + 1))
        # folded exchange (two collectives per step): the flags ride behind every rank's band in ONE all-gather; union = MAX
        # over the gathered copies; equal, load-balanced and tile-row-cyclic bands; twice, to cover buffer reuse
        for pf, idx_f in ((part, None), (pb, None), (pc, idx_c)):
            rows_f = np.array(pf.row_indices(), np.int64)
            if idx_f is None:
                idx_f = np.full_like(idx, -1)
                idx_f[:, rows_f] = idx[:, rows_f]
            vis_f = torch.from_numpy(oracle.visibility(idx_f, P).astype(np.uint8))
            oxf = OverlappedExchange(pf, 2, full.shape[-1], P, "cpu", fold=True)
            assert oxf.vrows >= 1 and oxf.visible.shape == (P,) and oxf.visible.dtype == torch.uint8
            for rep in range(2):
                oxf.image.copy_(full_t[:, rows_f] * (rep + 1))
                oxf.visible.copy_(vis_f)
                assert torch.equal(oxf.start(), vis), pf.describe()
                assert torch.equal(oxf.finish(), full_t * (rep + 1)), pf.describe()
        rs = oracle.backward_radius(sc["radii"], vis.numpy(), sc["first_idx"], sc["num_pts"], 3.0)
        gocc = g_full[..., 3].numpy()
        masked_c = np.zeros_like(gocc)
        masked_c[:, own] = gocc[:, own]
        g_c = torch.from_numpy(oracle.occ_backward_fast(sc["points"], sc["radii"], vis.numpy(), rs, masked_c,
                                                        sc["first_idx"], sc["num_pts"]))
        reduce_grads_(g_c, part=pc)
        assert np.allclose(g_c.numpy(), oracle.occ_backward_fast(sc["points"], sc["radii"], vis.numpy(), rs, gocc,
                                                                 sc["first_idx"], sc["num_pts"]), rtol=1e-4, atol=1e-4)
        masked = np.zeros_like(gocc)
        masked[:, r0:r1] = gocc[:, r0:r1]
        g_part = torch.from_numpy(oracle.occ_backward_fast(sc["points"], sc["radii"], vis.numpy(), rs, masked,
                                                           sc["first_idx"], sc["num_pts"]))
        g_masked = np.zeros_like(g_full.numpy())
        g_masked[:, r0:r1] = g_full[:, r0:r1].numpy()
        gf_part, _ = oracle.blend_backward(g_masked, idx, qv, sc["scaler"], P)
        gf_part = torch.from_numpy(gf_part)
        reduce_grads_(g_part, gf_part, part=part)
        g_all = oracle.occ_backward_fast(sc["points"], sc["radii"], vis.numpy(), rs, gocc, sc["first_idx"], sc["num_pts"])
        gf_all, _ = oracle.blend_backward(g_full.numpy(), idx, qv, sc["scaler"], P)
        assert np.allclose(g_part.numpy(), g_all, rtol=1e-4, atol=1e-4)
        assert np.allclose(gf_part.numpy(), gf_all, rtol=1e-4, atol=1e-5)
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S", [32, 37])
def test_row_partition_gloo_world2(tmp_path, S):
    port = 29600 + (os.getpid() % 200) + S
    mp.spawn(_worker, args=(2, port, S, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_row_partition_bounds():
    sys.path.insert(0, ROOT)
    from dss_amd.distributed import RowPartition
    for S, W in ((512, 8), (37, 2), (5, 8), (1024, 3)):
        rows = [RowPartition(S, W, r).rows for r in range(W)]
        assert rows[0][0] == 0 and rows[-1][1] == S or rows[-1][0] == rows[-1][1] == S
        cover = sum(b - a for a, b in rows)
        assert cover == S
        for (a, b), (c, d) in zip(rows[:-1], rows[1:]):
            assert b == c


def test_bounds_from_measured_times():
    """dss_amd.distributed.fitted_bounds / rebalanced_bounds (contiguous bands of the large multi-GPU workloads):
    per-rank times that follow  F + a * covered pixels + b * rows  give the model back from two different partitions, the
    fitted bands are level under that model, and a rebalancing step levels a cost profile the model does not know."""
    from dss_amd.distributed import RowPartition, balanced_bounds, fitted_bounds, rebalanced_bounds
    S, G = 1024, 8
    rows = np.arange(S)
    w = 60000.0 * np.exp(-((rows - 500) / 180.0) ** 2)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    model = lambda b: [700.0 + 2e-4 * (cum[b[i + 1]] - cum[b[i]]) + 1.2 * (b[i + 1] - b[i]) for i in range(G)]
    equal = [RowPartition(S, G, r).rows[0] for r in range(G)] + [S]
    by_occupancy = balanced_bounds(w, G)
    bounds, (F, a, b) = fitted_bounds(w, [(equal, model(equal)), (by_occupancy, model(by_occupancy))], G)
    assert abs(F - 700.0) < 1e-3 and abs(a - 2e-4) < 1e-9 and abs(b - 1.2) < 1e-6
    assert bounds[0] == 0 and bounds[-1] == S and all(y - x >= 8 and x % 8 == 0 for x, y in zip(bounds[:-1], bounds[1:]))
    t = model(bounds)
    assert max(t) - min(t) < 0.05 * (max(model(equal)) - min(model(equal))) + 2 * 8 * (2e-4 * w.max() + 1.2)
    # a cost per row that is neither occupancy nor constant: two rebalancing steps bring the spread down
    hidden = 1.0 + 4.0 * (rows > 700)
    hc = np.concatenate([[0.0], np.cumsum(hidden)])
    true = lambda bb: [300.0 + (hc[bb[i + 1]] - hc[bb[i]]) for i in range(G)]
    b1 = rebalanced_bounds(equal, true(equal), 300.0)
    b2 = rebalanced_bounds(b1, true(b1), 300.0)
    spread = lambda bb: max(true(bb)) - min(true(bb))
    assert spread(b2) < 0.25 * spread(equal)
    with pytest.raises(ValueError):
        balanced_bounds(w, 200)


def test_cyclic_row_partition_layout():
    """tile-row-cyclic partition: every image row has exactly one owner, band sizes agree with dss_band_rows' formula
    (ops.band_rows), the gather index inverts the rank-major concatenation of the bands"""
    sys.path.insert(0, ROOT)
    from dss_amd import ops
    from dss_amd.distributed import RowPartition
    for S, G in ((512, 8), (512, 2), (37, 2), (37, 4), (40, 4), (8, 4), (1024, 8)):
        parts = [RowPartition(S, G, g, cyclic=True) for g in range(G)]
        owned = sorted(r for p in parts for r in p.row_indices())
        assert owned == list(range(S))
        for g, p in enumerate(parts):
            ri = p.row_indices()
            assert len(ri) == p.n_rows == ops.band_rows(*p.rows) <= p.band
            assert all((r // 8) % G == g for r in ri) and ri == sorted(ri)
            # band row l <-> image row row0 + (l // 8) * 8 G + l % 8  (include/dss_hip.h)
            assert ri == [p.rows[0] + (l // 8) * 8 * G + l % 8 for l in range(len(ri))]
        if S % (8 * G) == 0:
            assert len({p.n_rows for p in parts}) == 1 and parts[0].band == S // G      # equal bands: no padding
        full = torch.arange(S * 3, dtype=torch.float32).reshape(1, S, 3)
        recv = torch.zeros((G * parts[0].band, 1, 3))
        for g, p in enumerate(parts):
            recv[g * p.band:g * p.band + p.n_rows] = p.slice(full).permute(1, 0, 2)
        assert torch.equal(recv[torch.tensor(parts[0].gather_index())].permute(1, 0, 2), full)
    with pytest.raises(ValueError):
        RowPartition(64, 3, 0, cyclic=True)       # power-of-two world sizes only (the kernels shift)
    assert RowPartition(64, 1, 0, cyclic=True).rows == (0, 64)      # one rank: the whole image, contiguous


def _band_loss_worker(rank, world, port, tmp, cyclic=False):
    """band_image_loss over gloo with the three HIP calls replaced by numpy stand-ins (same contracts): the all-reduce
    of the per-image sums and the autograd wiring are what is under test; the expected values come from the oracle on
    the full image."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from dss_amd import ops
    from dss_amd.distributed import RowPartition, band_image_loss
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(4)
        N, H, W = 2, 24, 20
        img = rng.random((N, H, W, 3)).astype(np.float32)
        rgba = rng.random((N, H, W, 4)).astype(np.float32)
        rgba[..., 3] = rng.random((N, H, W)) < 0.5
        mask = (rng.random((N, H, W)) < 0.5).astype(np.float32)
        lam = (0.7, 2.0)

        def owned(rows):   # image rows of (row0, row1[, cycle]) in band order
            c = rows[2] if len(rows) > 2 else 1
            return [r + i for r in range(rows[0], rows[1], 8 * c) for i in range(8) if r + i < rows[1]] if c > 1 \
                else list(range(rows[0], rows[1]))

        def band_sums(rgba_band, target_rgb, target_mask, rows):
            ri = owned(rows)
            a, t = rgba_band[..., 3].double(), target_mask.reshape(N, H, W)[:, ri].double()
            inside = (a != 0) & (t != 0)
            diff = (target_rgb[:, ri].double() - rgba_band[..., :3].double()).abs().sum(-1)
            s = torch.zeros(N + 1, 5, dtype=torch.float64)
            s[:N, 0] = inside.sum((1, 2)); s[:N, 1] = (diff * inside).sum((1, 2)); s[:N, 2] = (t - a).abs().sum((1, 2))
            s[:N, 3] = (a * t).sum((1, 2)); s[:N, 4] = (a + t - a * t).sum((1, 2))
            return s

        def from_sums(sums, image_size, l_rgb, l_sil):
            sums[N] = sums[:N].sum(0)
            iou = (1.0 - sums[:N, 3] / sums[:N, 4].clamp_min(1e-17)).mean()
            rgb = sums[N, 1] / sums[N, 0] if sums[N, 0] > 0 else torch.tensor(0.0, dtype=torch.float64)
            sil = sums[N, 2] / (N * image_size[0] * image_size[1]) + 0.01 * iou
            return torch.stack([l_rgb * rgb + l_sil * sil, l_rgb * rgb, l_sil * sil, iou]).float()

        def band_backward(rgba_band, target_rgb, target_mask, rows, l_rgb, l_sil, sums, grad_total=None):
            ri = owned(rows)
            a, t = rgba_band[..., 3].double(), target_mask.reshape(N, H, W)[:, ri].double()
            inside = ((a != 0) & (t != 0)).double()
            g = torch.zeros_like(rgba_band, dtype=torch.float64)
            g[..., :3] = l_rgb * torch.sign(rgba_band[..., :3].double() - target_rgb[:, ri].double()) * inside[..., None] / sums[N, 0]
            I, U = sums[:N, 3][:, None, None], sums[:N, 4][:, None, None]
            g[..., 3] = l_sil * (torch.sign(a - t) / (N * H * W) + 0.01 * (-(t * U - I * (1 - t)) / (U * U)) / N)
            return (g * (1.0 if grad_total is None else float(grad_total))).float()

        # the two-launch form band_image_loss uses: block partials (N, 64, 5) -> [all-reduce] -> gradient + losses
        def band_partials(rgba_band, target_rgb, target_mask, rows, band_targets=None, out=None):
            part = torch.zeros(N, 64, 5, dtype=torch.float64)
            s = band_sums(rgba_band, target_rgb, target_mask, rows)[:N]
            part[:, 0] = 0.25 * s          # spread over a few blocks: the consumer has to add them up
            part[:, 17] = 0.75 * s
            return part

        def band_backward_partials(rgba_band, target_rgb, target_mask, rows, l_rgb, l_sil, partials, grad_total=None,
                                   band_targets=None, want_sums=False):
            sums = torch.zeros(N + 1, 5, dtype=torch.float64)
            sums[:N] = partials.sum(1)
            losses = from_sums(sums, (H, W), l_rgb, l_sil)
            return band_backward(rgba_band, target_rgb, target_mask, rows, l_rgb, l_sil, sums, grad_total), losses

        ops.image_loss_band_partials, ops.image_loss_band_backward_partials = band_partials, band_backward_partials
        part = RowPartition(H, world, rank, cyclic=cyclic)
        ri = part.row_indices()
        band = torch.from_numpy(rgba[:, ri].copy()).requires_grad_(True)
        out = band_image_loss(band, torch.from_numpy(img), torch.from_numpy(mask), part, *lam)
        want, want_grad = oracle.image_loss(rgba, img, mask, *lam)
        assert abs(out["loss"].item() - want[0]) <= 1e-5 * abs(want[0]), (out["loss"].item(), want[0])
        assert abs(out["loss_dr_rgb"].item() - want[1]) <= 1e-5 * abs(want[1])
        (out["loss"] * 1.5).backward()
        assert np.allclose(band.grad.numpy(), 1.5 * want_grad[:, ri], rtol=1e-5, atol=1e-9)
        open(os.path.join(tmp, "band_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_band_image_loss_gloo_world2(tmp_path):
    port = 29850 + (os.getpid() % 100)
    mp.spawn(_band_loss_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "band_ok0").exists() and (tmp_path / "band_ok1").exists()


@pytest.mark.parametrize("world,S", [(4, 40), (4, 70), (8, 72)])
def test_row_partition_gloo_larger_worlds(tmp_path, world, S):
    """The same exchange steps at world sizes 4 and 8 (the first 8-GPU run should not be the first time a partition with
    more than two ranks is exercised): S = 40 / 70 leave ranks with one, two or three tile rows and a short last tile
    row, S = 72 at 8 ranks gives rank 0 two tile rows and every other rank one."""
    port = 29900 + (os.getpid() % 150) + S + world
    mp.spawn(_worker, args=(world, port, S, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(world))


def test_band_image_loss_gloo_world4(tmp_path):
    port = 29750 + (os.getpid() % 90)
    mp.spawn(_band_loss_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert all((tmp_path / ("band_ok%d" % r)).exists() for r in range(4))


def test_band_image_loss_gloo_world2_cyclic(tmp_path):
    """the layout `bench.py --gpus N` defaults to: `part.rows` is the triple (8 rank, S, G)"""
    port = 29650 + (os.getpid() % 90)
    mp.spawn(_band_loss_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    assert (tmp_path / "band_ok0").exists() and (tmp_path / "band_ok1").exists()


def test_gradient_exchange_choice_and_default_partition_without_a_process_group():
    """host logic of dss_amd.sharded: the automatic choice of the gradient exchange from the bytes on the critical path, and
    that no process group means no partition (the plain single-GPU path)"""
    sys.path.insert(0, ROOT)
    from dss_amd.sharded import choose_gradient_exchange, default_partition
    assert default_partition(512) is None
    # the metric's configuration (8 x 32,684 points, 512^2, band loss): 6.3 MB of partial sums beat an extra collective
    assert choose_gradient_exchange(8, 32684, 8 * 32684, 512, 3, 8, True, True) == "bucket"
    # BASELINE configs[3] (8 x 1M points, 1024^2): 192 MB of partial sums against 24 MB of world-space sums + a 4 MB plane
    assert choose_gradient_exchange(8, 10 ** 6, 8 * 10 ** 6, 1024, 3, 8, True, True) == "owner"
    # one camera (configs[4]): both forms reduce the same bytes, the owner form only adds the plane exchange
    assert choose_gradient_exchange(1, 4 * 10 ** 6, 4 * 10 ** 6, 2048, 3, 8, True, True) == "bucket"
    # a replicated loss hands the owner form the full gradient: no plane exchange, fewer bytes whenever cameras share a cloud
    assert choose_gradient_exchange(8, 32684, 8 * 32684, 512, 3, 8, False, False) == "owner"
    from dss_amd.distributed import agree_on_bounds
    assert agree_on_bounds([0, 8, 16]) == [0, 8, 16]          # no process group: handed back
    from dss_amd.distributed import fitted_bounds
    with pytest.raises(ValueError):
        fitted_bounds(torch.ones(16), [([0, 8, 16], [float("nan"), float("nan")])], 2)
