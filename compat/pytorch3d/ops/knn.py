"""knn_points / knn_gather with pytorch3d.ops.knn's interface.  GPU self-queries (the only kind the reference issues in
its training loop: cloud.py / losses.py / mathHelper.py call knn_points(p, p, lengths, lengths, K)) run on dss_amd's HIP
grid kNN; everything else is a chunked brute-force cdist + topk in torch."""
from collections import namedtuple

import torch

_KNN = namedtuple("KNN", "dists idx knn")


def _brute(p1, p2, lengths1, lengths2, K):
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    dists = p1.new_zeros((N, P1, K))
    idx = torch.zeros((N, P1, K), dtype=torch.int64, device=p1.device)
    for n in range(N):
        n1, n2 = int(lengths1[n]), int(lengths2[n])
        if n1 == 0 or n2 == 0:
            continue
        k = min(K, n2)
        a, b = p1[n, :n1], p2[n, :n2]
        step = max(1, (1 << 24) // max(n2, 1))
        for s in range(0, n1, step):
            q = a[s:s + step]
            d2 = (q * q).sum(-1, keepdim=True) - 2.0 * q @ b.t() + (b * b).sum(-1)[None]
            dk, ik = torch.topk(d2, k, dim=1, largest=False, sorted=True)
            # exact squared distances of the selected neighbours (the expansion above only ranks them)
            dk = ((q[:, None, :] - b[ik]) ** 2).sum(-1)
            order = torch.argsort(dk, dim=1, stable=True)
            e = s + q.shape[0]
            dists[n, s:e, :k] = torch.gather(dk, 1, order)
            idx[n, s:e, :k] = torch.gather(ik, 1, order)
    return dists, idx


def _self_query_on_gpu(p1, lengths1, K):
    """the padded batch handed over as it lies: cloud n = rows [n P1, n P1 + lengths1[n]) of the packed array, the rows behind a
    cloud's length belong to no cloud (the kernel zero-fills them).  No host value of `lengths1` is needed: slicing the clouds
    out with int(lengths1[n]) waited for the GPU once per cloud and call"""
    from dss_amd import ops  # the HIP path; raises if the library is missing
    N, P1, _ = p1.shape
    num = lengths1.to(torch.int64)
    first = torch.arange(N, device=p1.device, dtype=torch.int64) * P1
    packed = p1.detach().reshape(N * P1, 3).contiguous().float()
    d, i = ops.knn_points(packed, first, num, K)
    return d.view(N, P1, K).to(p1.dtype), i.view(N, P1, K)


def knn_points(p1, p2, lengths1=None, lengths2=None, K: int = 1, version: int = -1, return_nn: bool = False,
               return_sorted: bool = True):
    """K nearest neighbours in p2 of every point of p1: squared distances (N, P1, K) ascending, indices (N, P1, K) into
    p2, optionally the neighbours themselves.  Rows beyond lengths1 and columns beyond lengths2 are zero."""
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    p1 = p1.contiguous()
    p2 = p2.contiguous()
    N, P1, P2 = p1.shape[0], p1.shape[1], p2.shape[1]
    both_full = lengths1 is None and lengths2 is None and P1 == P2
    if lengths1 is None:
        lengths1 = torch.full((N,), P1, dtype=torch.int64, device=p1.device)
    if lengths2 is None:
        lengths2 = lengths1 if both_full else torch.full((N,), P2, dtype=torch.int64, device=p1.device)
    # (comparisons that ask the device only when the cheap identity tests do not settle the question)
    same = p1.is_cuda and p1.shape == p2.shape and p1.shape[2] == 3 and K <= 64 \
        and (p1.data_ptr() == p2.data_ptr() or torch.equal(p1, p2)) \
        and (lengths1 is lengths2 or lengths1.data_ptr() == lengths2.data_ptr() or torch.equal(lengths1, lengths2))
    with torch.no_grad():
        if same:
            _, idx = _self_query_on_gpu(p1, lengths1, K)
        else:
            _, idx = _brute(p1.detach(), p2.detach(), lengths1, lengths2, K)
    # distances recomputed from the indices so that they are differentiable w.r.t. p1 and p2 (as in pytorch3d)
    nn = knn_gather(p2, idx, lengths2)
    dists = ((p1[:, :, None, :] - nn) ** 2).sum(-1)
    k_valid = torch.arange(K, device=p1.device)[None, None, :] < lengths2.view(-1, 1, 1)
    row_valid = torch.arange(P1, device=p1.device)[None, :, None] < lengths1.view(-1, 1, 1)
    dists = torch.where(k_valid & row_valid, dists, torch.zeros_like(dists))
    idx = torch.where(k_valid & row_valid, idx, torch.zeros_like(idx))
    return _KNN(dists=dists, idx=idx, knn=nn if return_nn else None)


def knn_gather(x, idx, lengths=None):
    """x (N, M, U), idx (N, L, K) -> (N, L, K, U); neighbours k >= lengths[n] are zero"""
    N, M, U = x.shape
    _N, L, K = idx.shape
    if N != _N:
        raise ValueError("x and idx must have same batch dimension.")
    if lengths is None:
        lengths = torch.full((x.shape[0],), M, dtype=torch.int64, device=x.device)
    idx_expanded = idx[:, :, :, None].expand(-1, -1, -1, U)
    x_out = x[:, :, None].expand(-1, -1, K, -1).gather(1, idx_expanded)
    needs_mask = True if x.is_cuda else bool(lengths.min() < K)   # (on a GPU the question costs more than the mask: a host sync)
    if needs_mask:
        mask = lengths[:, None] <= torch.arange(K, device=x.device)[None]
        mask = mask[:, None].expand(-1, L, -1)
        mask = mask[:, :, :, None].expand(-1, -1, -1, U)
        x_out = x_out.masked_fill(mask, 0.0)
    return x_out
