"""GPU: the fused image loss of the training iteration (Trainer.calc_dr_loss, trainer.py:332-372) through the C ABI,
against the golden vectors produced by the reference method itself with autograd
(tests/golden/make_golden_image_loss.py) and against the oracle at full image sizes.  Float reductions: loss values
to 1e-5 relative, gradients elementwise to 1e-5 relative (they are signs times exact scale factors)."""
import os

import numpy as np
import pytest
import torch

import oracle
from dss_amd import ops
from dss_amd.losses import calc_dr_loss

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_image_loss.npz")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_matches_reference_trainer_golden(tag):
    z = np.load(GOLD)
    rgba = _t(np.concatenate([z[tag + "_pred"], z[tag + "_mask_pred"][..., None]], -1)).requires_grad_(True)
    lam_rgb, lam_sil = (float(v) for v in z[tag + "_lambdas"])
    # the target arrives as the permuted view of an NCHW tensor, like trainer.py:306
    img_nchw = _t(z[tag + "_img"].transpose(0, 3, 1, 2))
    out = calc_dr_loss(rgba, img_nchw.permute(0, 2, 3, 1), _t(z[tag + "_mask"])[:, None], lam_rgb, lam_sil)
    assert abs(out["loss"].item() - z[tag + "_loss"]) <= 1e-5 * abs(z[tag + "_loss"])
    assert abs(out["loss_dr_rgb"].item() - z[tag + "_loss_rgb"]) <= 1e-5 * max(abs(z[tag + "_loss_rgb"]), 1e-6)
    assert abs(out["loss_dr_silhouette"].item() - z[tag + "_loss_sil"]) <= 1e-5 * abs(z[tag + "_loss_sil"])
    out["loss"].backward()
    g = rgba.grad.cpu().numpy()
    assert np.allclose(g[..., :3], z[tag + "_grad_pred"], rtol=1e-5, atol=1e-9)
    assert np.allclose(g[..., 3], z[tag + "_grad_mask_pred"], rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("N,H,W", [(1, 512, 512), (8, 1024, 1024), (3, 37, 53)])
def test_matches_oracle_at_image_sizes(N, H, W):
    rng = np.random.default_rng(N * 1000 + H)
    img = rng.random((N, H, W, 3)).astype(np.float32)
    rgba = rng.random((N, H, W, 4)).astype(np.float32)
    rgba[..., 3] = rng.random((N, H, W)) < 0.4          # hard occupancy like the renderer's alpha
    mask = (rng.random((N, H, W)) < 0.5).astype(np.float32)
    losses, sums = ops.image_loss_forward(_t(rgba), _t(img), _t(mask), 1.0, 1.0)
    up = torch.tensor([0.37], device=DEV)
    grad = ops.image_loss_backward(_t(rgba), _t(img), _t(mask), 1.0, 1.0, sums, grad_total=up)
    lo, go = oracle.image_loss(rgba, img, mask, 1.0, 1.0)
    assert np.allclose(losses.cpu().numpy(), lo, rtol=1e-5)
    assert np.allclose(grad.cpu().numpy(), 0.37 * go, rtol=1e-5, atol=1e-12)
    s = sums.cpu().numpy()
    inside = (mask != 0) & (rgba[..., 3] != 0)
    assert np.array_equal(s[:N, 0], inside.reshape(N, -1).sum(1).astype(np.float64))     # counts are exact
    assert s[N, 0] == inside.sum() and np.allclose(s[N], s[:N].sum(0))
    # bit-reproducible: fixed-order partial sums
    losses2, sums2 = ops.image_loss_forward(_t(rgba), _t(img), _t(mask), 1.0, 1.0)
    assert torch.equal(losses, losses2) and torch.equal(sums, sums2)


def test_bad_arguments_fail_loudly():
    rgba = torch.rand(1, 8, 8, 4, device=DEV)
    img = torch.rand(1, 8, 8, 3, device=DEV)
    mask = torch.rand(1, 8, 8, device=DEV)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.image_loss_forward(rgba, img.cpu(), mask, 1.0, 1.0)
    with pytest.raises(RuntimeError, match=r"\(N,H,W,3\)"):
        ops.image_loss_forward(rgba, img.permute(0, 3, 1, 2), mask, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="GPU tensors"):
        ops.image_loss_forward(rgba.cpu(), img, mask, 1.0, 1.0)
    _, sums = ops.image_loss_forward(rgba, img, mask, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="sums"):
        ops.image_loss_backward(rgba, img, mask, 1.0, 1.0, sums[:1])


def test_degenerate_images():
    """All-empty prediction and target (union = 0: IoU term 1 - 0/eps = 1, no rgb term) and a single pixel."""
    for N, H, W in ((2, 16, 16), (1, 1, 1)):
        rgba = torch.rand(N, H, W, 4, device=DEV)
        rgba[..., 3] = 0
        img = torch.rand(N, H, W, 3, device=DEV)
        mask = torch.zeros(N, H, W, device=DEV)
        losses, sums = ops.image_loss_forward(rgba, img, mask, 1.0, 1.0)
        grad = ops.image_loss_backward(rgba, img, mask, 1.0, 1.0, sums)
        lo, go = oracle.image_loss(rgba.cpu().numpy(), img.cpu().numpy(), mask.cpu().numpy(), 1.0, 1.0)
        assert np.allclose(losses.cpu().numpy(), lo, rtol=1e-6) and abs(lo[0] - 0.01) < 1e-7 and lo[1] == 0
        assert np.allclose(grad.cpu().numpy(), go, rtol=1e-5, atol=1e-12) and torch.isfinite(grad).all()


@pytest.mark.parametrize("bounds", [(0, 64, 128), (0, 40, 41, 128), (0, 0, 128)])
def test_row_bands_reproduce_the_full_image_loss(bounds):
    """Multi-GPU form: per-band sums added up (what the all-reduce does) give the full-image loss, and each
    band's gradient is the matching slice of the full-image gradient.  Bands of 64+64, 40+1+87 and an empty band."""
    from dss_amd.distributed import RowPartition, band_image_loss
    rng = np.random.default_rng(21)
    N, H, W = 3, 128, 96
    img = _t(rng.random((N, 3, H, W)).astype(np.float32)).permute(0, 2, 3, 1)          # NCHW view, like the trainer
    rgba = rng.random((N, H, W, 4)).astype(np.float32)
    rgba[..., 3] = rng.random((N, H, W)) < 0.4
    rgba = _t(rgba)
    mask = _t((rng.random((N, H, W)) < 0.5).astype(np.float32))
    losses, sums = ops.image_loss_forward(rgba, img, mask, 0.7, 2.0)
    up = torch.tensor([1.3], device=DEV)
    grad = ops.image_loss_backward(rgba, img, mask, 0.7, 2.0, sums, grad_total=up)

    G = len(bounds) - 1
    band_sums = [ops.image_loss_band_sums(rgba[:, bounds[g]:bounds[g + 1]].contiguous(), img, mask, (bounds[g], bounds[g + 1]))
                 for g in range(G)]
    reduced = torch.stack(band_sums).sum(0)                                               # the all-reduce
    assert torch.allclose(reduced[:N], sums[:N], rtol=1e-6, atol=0) and torch.equal(reduced[:N, 0], sums[:N, 0])
    losses_b = ops.image_loss_from_sums(reduced, (H, W), 0.7, 2.0)
    assert torch.allclose(losses_b, losses, rtol=1e-6)
    for g in range(G):
        r0, r1 = bounds[g], bounds[g + 1]
        gb = ops.image_loss_band_backward(rgba[:, r0:r1].contiguous(), img, mask, (r0, r1), 0.7, 2.0, reduced, grad_total=up)
        assert tuple(gb.shape) == (N, r1 - r0, W, 4)
        assert torch.allclose(gb, grad[:, r0:r1], rtol=1e-6, atol=1e-12)
    # the two-launch form (what band_image_loss and bench.py's multi-GPU step use): the block partials are what the ranks
    # all-reduce; gradient, losses and sums come out of ONE launch; precomputed band targets give the same bits
    parts_ = [ops.image_loss_band_partials(rgba[:, bounds[g]:bounds[g + 1]].contiguous(), img, mask, (bounds[g], bounds[g + 1]))
              for g in range(G)]
    red = torch.stack(parts_).sum(0)                                                      # the all-reduce
    for g in range(G):
        r0, r1 = bounds[g], bounds[g + 1]
        band = rgba[:, r0:r1].contiguous()
        gb, lb, sb = ops.image_loss_band_backward_partials(band, img, mask, (r0, r1), 0.7, 2.0, red, grad_total=up, want_sums=True)
        assert torch.allclose(gb, grad[:, r0:r1], rtol=1e-6, atol=1e-12) and torch.allclose(lb, losses, rtol=1e-6)
        # (the per-thread sums are fp32 over another partition of the pixels: counts exact, the real-valued terms to fp32)
        assert torch.allclose(sb, sums, rtol=1e-6) and torch.equal(sb[:, 0], sums[:, 0])
        if r1 > r0:
            bt = ops.band_targets(img, mask, (r0, r1))
            gb2, lb2 = ops.image_loss_band_backward_partials(band, img, mask, (r0, r1), 0.7, 2.0, red, grad_total=up, band_targets=bt)
            assert torch.equal(gb2, gb) and torch.equal(lb2, lb)
    # the autograd wrapper on a single rank (no process group): the band IS the image
    part = RowPartition(H, 1, 0)
    leaf = rgba.clone().requires_grad_(True)
    out = band_image_loss(leaf, img, mask[:, None], part, 0.7, 2.0)
    assert torch.allclose(out["loss"], losses[0], rtol=1e-6)
    (out["loss"] * 1.3).backward()
    assert torch.allclose(leaf.grad, grad, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("G,H", [(2, 128), (4, 128), (4, 72)])
def test_tile_row_cyclic_bands_reproduce_the_full_image_loss(G, H):
    """`bench.py --gpus N` partitions tile-row-CYCLICALLY (`RowPartition(cyclic=True).rows` = (8 rank, S, G));
    the band loss must take that triple: per-band sums add up to the full-image sums, every band's gradient is the
    owned rows of the full-image gradient (H = 72 at G = 4: bands of 24 / 16 / 16 / 16 rows)."""
    from dss_amd.distributed import RowPartition
    rng = np.random.default_rng(33)
    N, W = 2, 96
    img = _t(rng.random((N, 3, H, W)).astype(np.float32)).permute(0, 2, 3, 1)
    rgba = rng.random((N, H, W, 4)).astype(np.float32)
    rgba[..., 3] = rng.random((N, H, W)) < 0.4
    rgba = _t(rgba)
    mask = _t((rng.random((N, H, W)) < 0.5).astype(np.float32))
    losses, sums = ops.image_loss_forward(rgba, img, mask, 0.7, 2.0)
    up = torch.tensor([1.3], device=DEV)
    grad = ops.image_loss_backward(rgba, img, mask, 0.7, 2.0, sums, grad_total=up)
    parts = [RowPartition(H, G, g, cyclic=True) for g in range(G)]
    bands = [p.slice(rgba).contiguous() for p in parts]
    reduced = torch.stack([ops.image_loss_band_sums(b, img, mask, p.rows) for b, p in zip(bands, parts)]).sum(0)
    assert torch.allclose(reduced[:N], sums[:N], rtol=1e-6, atol=0) and torch.equal(reduced[:N, 0], sums[:N, 0])
    assert torch.allclose(ops.image_loss_from_sums(reduced, (H, W), 0.7, 2.0), losses, rtol=1e-6)
    red = torch.stack([ops.image_loss_band_partials(b, img, mask, p.rows) for b, p in zip(bands, parts)]).sum(0)
    for b, p in zip(bands, parts):
        gb = ops.image_loss_band_backward(b, img, mask, p.rows, 0.7, 2.0, reduced, grad_total=up)
        assert tuple(gb.shape) == (N, p.n_rows, W, 4)
        assert torch.allclose(gb, p.slice(grad), rtol=1e-6, atol=1e-12)
        # two-launch form, with the band's targets gathered once (tile-row-cyclic: contiguous copies of the owned rows)
        bt = tuple(x.contiguous() for x in ops.band_targets(img, mask, p.rows))
        gb2, lb2 = ops.image_loss_band_backward_partials(b, img, mask, p.rows, 0.7, 2.0, red, grad_total=up, band_targets=bt)
        assert torch.allclose(gb2, p.slice(grad), rtol=1e-6, atol=1e-12) and torch.allclose(lb2, losses, rtol=1e-6)
        # the band where the multi-GPU forward leaves it: a (row, camera, col, channel) send buffer seen through its strided
        # (N, rows, W, 4) view -- no copy --, and the alpha channel of the gradient written once more into the (row, camera, col)
        # send buffer of the owner form's exchange: same bits as the dense call
        send = torch.zeros((p.n_rows + 3, N, W, 4), device=DEV)
        view = send[:p.n_rows].permute(1, 0, 2, 3)
        view.copy_(b)
        assert not view.is_contiguous()
        assert torch.equal(ops.image_loss_band_partials(view, img, mask, p.rows, band_targets=bt),
                           ops.image_loss_band_partials(b, img, mask, p.rows, band_targets=bt))
        asend = torch.full((p.n_rows + 3, N, W), 7.0, device=DEV)
        aview = asend[:p.n_rows].permute(1, 0, 2)
        gb3, lb3 = ops.image_loss_band_backward_partials(view, img, mask, p.rows, 0.7, 2.0, red, grad_total=up, band_targets=bt,
                                                         alpha_out=aview)
        assert torch.equal(gb3, gb2) and torch.equal(lb3, lb2) and gb3.is_contiguous()
        assert torch.equal(aview, gb2[..., 3]) and bool((asend[p.n_rows:] == 7.0).all())
