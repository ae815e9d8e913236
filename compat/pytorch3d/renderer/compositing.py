"""Point compositing functions with pytorch3d.renderer.compositing's interface, as differentiable torch expressions.
Arguments: pointsidx (N, K, H, W) int64 indices into the packed features (-1 = empty), alphas (N, K, H, W),
pt_clds (C, P) features of the packed points.  Output (N, C, H, W)."""
import torch

kEpsilon = 1e-4


def _gather(pointsidx, pt_clds):
    valid = pointsidx >= 0
    idx = pointsidx.clamp(min=0).long()
    feats = pt_clds[:, idx]                      # (C, N, K, H, W)
    return feats * valid.to(feats.dtype)[None], valid


def weighted_sum(pointsidx, alphas, pt_clds) -> torch.Tensor:
    feats, valid = _gather(pointsidx, pt_clds)
    a = alphas * valid.to(alphas.dtype)
    return (feats * a[None]).sum(2).permute(1, 0, 2, 3)


def norm_weighted_sum(pointsidx, alphas, pt_clds) -> torch.Tensor:
    feats, valid = _gather(pointsidx, pt_clds)
    a = alphas * valid.to(alphas.dtype)
    denom = a.sum(1, keepdim=True).clamp(min=kEpsilon)
    return (feats * (a / denom)[None]).sum(2).permute(1, 0, 2, 3)


def alpha_composite(pointsidx, alphas, pt_clds) -> torch.Tensor:
    feats, valid = _gather(pointsidx, pt_clds)
    a = alphas * valid.to(alphas.dtype)
    transmittance = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1.0 - a[:, :-1]], 1), 1)
    return (feats * (a * transmittance)[None]).sum(2).permute(1, 0, 2, 3)
