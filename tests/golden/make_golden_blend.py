"""Generates tests/golden/ref_blend.npz: the reference's blend (weights, compositor, RGBA assembly), EXECUTED.

What runs (in this container, CPU only), all from where it lies under /root/reference:
  * `gather_with_neg_idx` exactly as `SurfaceSplatting.forward` calls it (DSS/core/rasterizer.py:631-633,
    DSS/utils/__init__.py:172-185)                                                  -> fragments.scaler
  * the UNMODIFIED `SurfaceSplattingRenderer.forward` (DSS/core/renderer.py:36-82): weights = exp(-Q/2) * scaler,
    permutes, compositor call, RGBA assembly, on the fragments of the compiled reference rasterizer
  * `weighted_sum` (renderer.py:59-65, `compositor=None`) = the reference's own copy of the weighted-sum compositor
    kernels, DSS/csrc/weighted_sum.cu:38-134, UNMODIFIED, host-compiled by oracle/ref_cuda_host.cpp; forward and backward
    (launch shape of weighted_sum.cu:160-161 / :206-207)
  * autograd of torch through all of it (d(sum(img * grad_out)) / d features).
`NormWeightedCompositor` (renderer.py:67-72) is pytorch3d (pinned 0.2.5 README.md:29-39 / 0.4.0 environment.yml:102;
NOT under /root/reference).  Its published algorithm (pytorch3d/csrc/compositing/norm_weighted_sum.cu: cum_alpha = sum of
the alphas of the fragments with idx >= 0, clamped below at kEpsilon = 1e-4; result = sum features * alpha / cum_alpha;
backward to the features alpha / cum_alpha * grad) is restated here as `weighted_sum` on normalised alphas, i.e. on top of
the executed reference kernel: vectors `norm_*`.

    python tests/golden/make_golden_blend.py
"""
import ctypes
import importlib
import os
import sys

sys.dont_write_bytecode = True  # the reference checkout is read-only: importing it must not leave __pycache__ there
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden_setup as mgs  # noqa: E402  (stub importer + the reference rasterizer module)

HOST = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_cuda_host.so"))
_p = lambda t: ctypes.c_void_p(t.data_ptr())
_i64 = ctypes.c_int64


class _WeightedSum(torch.autograd.Function):
    """pytorch3d's `_CompositeWeightedSumPoints` wiring around the reference's kernels (weighted_sum.cu)."""

    @staticmethod
    def forward(ctx, features, alphas, points_idx):
        features, alphas, points_idx = features.contiguous().float(), alphas.contiguous().float(), points_idx.contiguous().long()
        N, K, H, W = points_idx.shape
        C, P = features.shape
        out = torch.empty((N, C, H, W), dtype=torch.float32)
        HOST.ref_weighted_sum_forward(_p(features), _p(alphas), _p(points_idx), _i64(N), _i64(K), _i64(H), _i64(W), _i64(C),
                                      _i64(P), _p(out))
        ctx.save_for_backward(features, alphas, points_idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, alphas, points_idx = ctx.saved_tensors
        N, K, H, W = points_idx.shape
        C, P = features.shape
        gf, ga = torch.empty_like(features), torch.empty_like(alphas)
        grad_out = grad_out.contiguous().float()   # (named: a temporary would be freed before the call reads it)
        HOST.ref_weighted_sum_backward(_p(grad_out), _p(features), _p(alphas), _p(points_idx), _i64(N),
                                       _i64(K), _i64(H), _i64(W), _i64(C), _i64(P), _p(gf), _p(ga))
        return gf, ga, None


def weighted_sum(pointsidx, alphas, pt_clds, **kwargs):
    """pytorch3d.renderer.compositing.weighted_sum(pointsidx, alphas, pt_clds) (argument order of renderer.py:62-64)."""
    return _WeightedSum.apply(pt_clds, alphas, pointsidx)


class NormWeightedCompositor(torch.nn.Module):
    """pytorch3d's NormWeightedCompositor restated from its published algorithm on top of the executed weighted-sum kernel."""

    def forward(self, fragments, alphas, ptclds, **kwargs):
        valid = (fragments >= 0).to(alphas.dtype)
        cum = (alphas * valid).sum(dim=1, keepdim=True).clamp(min=1e-4)   # kEpsilon
        return weighted_sum(fragments, alphas / cum, ptclds)


class PointsRenderer(torch.nn.Module):
    def __init__(self, rasterizer, compositor):
        super().__init__()
        self.rasterizer, self.compositor = rasterizer, compositor


import pytorch3d.renderer as p3r  # noqa: E402  (the stub)
import pytorch3d.renderer.compositing as p3c  # noqa: E402
p3r.PointsRenderer, p3r.NormWeightedCompositor, p3c.weighted_sum = PointsRenderer, NormWeightedCompositor, weighted_sum
ref_renderer = importlib.import_module("DSS.core.renderer")     # the UNMODIFIED reference module
ref_utils = importlib.import_module("DSS.utils")
PointFragments = mgs.ref_rast.PointFragments


def run(z, compositor, grad_out):
    t = lambda k: torch.from_numpy(np.ascontiguousarray(z[k]))
    idx, qv, occ = t("ref_idx"), t("ref_qvalue"), t("ref_occ")
    # rasterizer.py:631-633
    frag_scaler = ref_utils.gather_with_neg_idx(t("scaler"), 0, idx.view(-1).long()).view_as(qv)
    fragments = PointFragments(idx=idx, zbuf=t("ref_zbuf"), qvalue=qv, scaler=frag_scaler, occupancy=occ)
    feats = t("colors").clone().requires_grad_(True)
    cloud = types.SimpleNamespace(isempty=lambda: False, features_packed=lambda: feats)
    # the rasterizer stage is replaced by the committed output of the compiled reference rasterizer (renderer.py:47-50
    # takes `(fragments, point_clouds)` from it; the `fragments=` keyword cannot be used with a compositor because
    # renderer.py:67-72 forwards **kwargs into `compositor(fragments, ...)`)
    class _Rasterizer:
        cameras = None

        def __call__(self, point_clouds, **kwargs):
            return fragments, point_clouds

    renderer = ref_renderer.SurfaceSplattingRenderer(_Rasterizer(), compositor)
    img = renderer(cloud)
    (img * torch.from_numpy(grad_out)).sum().backward()
    weights = (torch.exp(-0.5 * qv) * frag_scaler).permute(0, 3, 1, 2)   # renderer.py:53-54 (for the record only)
    return img.detach().numpy(), feats.grad.numpy(), weights.numpy(), frag_scaler.numpy()


def main():
    out = {}
    for name in ("ref_random64x2", "ref_teapot256", "ref_ties32"):
        z = np.load(os.path.join(HERE, name + ".npz"))
        N, S = z["ref_occ"].shape[0], int(z["S"])
        grad_out = np.random.default_rng(11).standard_normal((N, S, S, 4)).astype(np.float32)
        out[name + "_grad_out"] = grad_out
        img, gf, w, fs = run(z, None, grad_out)
        out[name + "_ws_image"], out[name + "_ws_grad_features"] = img, gf
        out[name + "_weights"], out[name + "_frag_scaler"] = w, fs
        img, gf, _, _ = run(z, NormWeightedCompositor(), grad_out)
        out[name + "_norm_image"], out[name + "_norm_grad_features"] = img, gf
    np.savez_compressed(os.path.join(HERE, "ref_blend.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()))


if __name__ == "__main__":
    main()
