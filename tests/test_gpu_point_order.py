"""GPU: the renderer-owned cached point order (include/dss_hip.h DSS_WS_ORDER_SAVE / DSS_WS_ORDER_REUSE).
Above 2M points the binning runs in screen-cell order; a call can leave that order in the workspace and later calls bin
through it instead of sorting.  The outputs must not depend on the order in any bit -- also when the order is STALE: saved
under another camera, with another set of culled points (back-face culling on), or for moved points."""
import numpy as np
import pytest
import torch

import scenes
from dss_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K, THR, CUTOFF, SIGMA = 5, 0.05, 1.0, 1.0
P, S = 2_200_037, 512   # above SORT_MIN_P (raster_forward.hip); not a multiple of 64: the last wavefront is ragged


def _inputs(azim, seed=0, shift=0.0):
    pts, nrm, col = scenes.synthetic_cloud(P, seed=seed)
    if shift:
        pts = (pts + shift * np.random.default_rng(5).standard_normal(pts.shape)).astype(np.float32)
    M, V, _ = scenes.camera_matrices(2.0, 20.0, [azim])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    one = lambda v: torch.full((1,), v, device=DEV)
    first = torch.zeros(1, device=DEV, dtype=torch.int64)
    num = torch.full((1,), P, device=DEV, dtype=torch.int64)
    return (t(pts), t(nrm), one(2e-5), t(M), t(V), one(0.1), one(100.0), first, num, t(col))


def _render(inp, **kw):
    return ops.render_forward(*inp, S, K, CUTOFF, THR, SIGMA, True, False, **kw)   # back-face culling ON


def _same(a, b):
    for key in ("idx", "zbuf", "qvalue", "occupancy", "visible", "valid", "image", "wsum", "pts_screen", "radii",
                "ellipse_params", "scaler", "cutoff_threshold"):
        assert torch.equal(a[key], b[key]), key


def test_cached_point_order_is_invisible_in_the_outputs():
    a, b = _inputs(0.0), _inputs(135.0)
    ref_a, ref_b = _render(a, workspace_state=0), _render(b, workspace_state=0)   # (own workspace, sorts for itself)
    assert 0.2 < float(ref_a["valid"].float().mean()) < 0.8, "back-face culling is meant to drop about half of the points"
    assert not torch.equal(ref_a["valid"], ref_b["valid"])
    moved = _inputs(135.0, shift=0.02)
    ref_m = _render(moved, workspace_state=0)
    # call 1 saves the order of camera A; calls 2-4 reuse it: same camera, another camera (other culled set), moved points
    _same(_render(a, order_refresh=4), ref_a)
    _same(_render(a, order_refresh=4), ref_a)
    _same(_render(b, order_refresh=4), ref_b)
    _same(_render(moved, order_refresh=4), ref_m)
    # call 5 refreshes (saves the order of the moved cloud), call 6 reuses it for camera A
    _same(_render(moved, order_refresh=4), ref_m)
    _same(_render(a, order_refresh=4), ref_a)
    # a call WITHOUT the option on the same workspace in between (another renderer of the same shape): it sorts for itself, the
    # library forgets the saved order, and the next call with the option saves again instead of failing
    _same(_render(b), ref_b)
    _same(_render(a, order_refresh=4), ref_a)
    _same(_render(b, order_refresh=4), ref_b)
    # and the gradients through the fragments of a reused order
    f = _render(b, order_refresh=4)
    g = torch.randn(1, S, S, 4, device=DEV, generator=torch.Generator(DEV).manual_seed(3))
    got = ops.render_backward(g, f["idx"], f["qvalue"], f["wsum"], f["scaler"], f["pts_screen"], f["radii"], f["visible"],
                              b[7], b[8], 5.0, 0.05)
    want = ops.render_backward(g, ref_b["idx"], ref_b["qvalue"], ref_b["wsum"], ref_b["scaler"], ref_b["pts_screen"],
                               ref_b["radii"], ref_b["visible"], b[7], b[8], 5.0, 0.05)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_reuse_without_a_saved_order_is_refused():
    from dss_amd import _lib
    a = _inputs(0.0)
    _render(a, workspace_state=0)   # a plain call on the buffer of the non-clean states: no order saved there
    with pytest.raises(RuntimeError, match="DSS_WS_ORDER_REUSE"):
        _render(a, workspace_state=0 | _lib.WS_ORDER_REUSE)
    # a save on that buffer, then the reuse is accepted; a plain call overwrites the order and the reuse is refused again
    ref = _render(a, workspace_state=0 | _lib.WS_ORDER_SAVE)
    _same(_render(a, workspace_state=0 | _lib.WS_ORDER_REUSE), ref)
    _render(a, workspace_state=0)
    with pytest.raises(RuntimeError, match="DSS_WS_ORDER_REUSE"):
        _render(a, workspace_state=0 | _lib.WS_ORDER_REUSE)


def test_order_flags_are_ignored_by_the_direct_binning():
    from dss_amd import _lib
    pts, nrm, col = scenes.synthetic_cloud(50_000, seed=1)
    M, V, _ = scenes.camera_matrices(2.0, 20.0, [30.0])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    one = lambda v: torch.full((1,), v, device=DEV)
    inp = (t(pts), t(nrm), one(1e-3), t(M), t(V), one(0.1), one(100.0), torch.zeros(1, device=DEV, dtype=torch.int64),
           torch.full((1,), 50_000, device=DEV, dtype=torch.int64), t(col))
    ref = ops.render_forward(*inp, 256, K, CUTOFF, THR, SIGMA, False, False, workspace_state=0)
    for flag in (_lib.WS_ORDER_REUSE, _lib.WS_ORDER_SAVE):
        got = ops.render_forward(*inp, 256, K, CUTOFF, THR, SIGMA, False, False, workspace_state=0 | flag)
        assert torch.equal(got["idx"], ref["idx"]) and torch.equal(got["image"], ref["image"])
    with pytest.raises(RuntimeError, match="workspace_state"):
        ops.render_forward(*inp, 256, K, CUTOFF, THR, SIGMA, False, False, workspace_state=0x80)
