"""Generates tests/golden/ref_losses.npz by running the REFERENCE's own point-cloud regularisers
(`ProjectionLoss`, `RepulsionLoss`, /root/reference/DSS/training/losses.py:145-459) in this container, with
autograd providing the gradients with respect to the points.

Stubbing is shared with make_golden_setup.py (importing it installs the auto-stubs for the absent third-party
modules and the brute-force `knn_points` stand-in).  Added here:
  - `pytorch3d.ops.knn_gather`   (gather neighbour rows)          -> torch.gather stand-in
  - `pytorch3d.ops.knn._KNN`     (namedtuple dists/idx/knn)       -> the same namedtuple as the knn stand-in
The clouds are minimal duck-typed objects with the handful of Pointclouds accessors the losses touch
(points_padded, normals_padded, num_points_per_cloud, cloud_to_packed_first_idx, get_bounding_boxes, clone,
update_normals_).  The reference source is imported from where it lies; nothing is copied.

    python tests/golden/make_golden_losses.py
"""
import importlib
import os
import types

import numpy as np
import torch

import make_golden_setup as base  # noqa: F401  (installs the stubs, puts /root/reference on sys.path)
import pytorch3d.ops as ops3d
import pytorch3d.ops.knn as ops3d_knn

HERE = os.path.dirname(os.path.abspath(__file__))


def _knn_gather(x, idx, lengths=None):
    N, P, U = x.shape
    K = idx.shape[2]
    out = x[:, :, None, :].expand(N, P, K, U).gather(1, idx[:, :, :, None].expand(N, P, K, U))
    return out


def _knn_points_padded(p1, p2, lengths1=None, lengths2=None, K=1, return_nn=False, **kw):
    """pytorch3d zero-fills the rows of padded queries and the columns beyond a short cloud's size."""
    r = base._knn_points(p1, p2, lengths1, lengths2, K, return_nn)
    dists, idx, nn = r.dists.clone(), r.idx.clone(), r.knn
    for b in range(p1.shape[0]):
        l1 = int(lengths1[b]) if lengths1 is not None else p1.shape[1]
        l2 = int(lengths2[b]) if lengths2 is not None else p2.shape[1]
        dists[b, l1:] = 0
        idx[b, l1:] = 0
        dists[b, :, l2:] = 0
        idx[b, :, l2:] = 0
    if return_nn:
        nn = torch.stack([p2[b][idx[b]] for b in range(p2.shape[0])], 0)
    return base._KNN(dists, idx, nn)


ops3d.knn_points = _knn_points_padded
ops3d.knn_gather = _knn_gather
ops3d_knn._KNN = base._KNN
ref_losses = importlib.import_module("DSS.training.losses")  # the UNMODIFIED reference module

import scenes  # noqa: E402


class _Clouds:
    """The Pointclouds accessors losses.py uses, over padded tensors."""

    def __init__(self, points, normals, lengths):
        self._p, self._n, self._len = points, normals, lengths

    def __len__(self):
        return self._p.shape[0]

    def points_padded(self):
        return self._p

    def normals_padded(self):
        return self._n

    def num_points_per_cloud(self):
        return self._len

    def cloud_to_packed_first_idx(self):
        return torch.cumsum(self._len, 0) - self._len

    def get_bounding_boxes(self):  # (N,3,2) min / max over the valid points
        lo = torch.stack([self._p[b, : int(l)].min(0).values for b, l in enumerate(self._len)])
        hi = torch.stack([self._p[b, : int(l)].max(0).values for b, l in enumerate(self._len)])
        return torch.stack([lo, hi], -1)

    def clone(self):
        return _Clouds(self._p, self._n.clone(), self._len)

    def update_normals_(self, packed):
        first = self.cloud_to_packed_first_idx()
        n = torch.zeros_like(self._n)
        for b, l in enumerate(self._len):
            n[b, : int(l)] = packed[int(first[b]): int(first[b]) + int(l)]
        self._n = n


def main():
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    pts, nrm = scenes.load_cloud("teapot")
    pts = scenes.normalize_unit_sphere(pts)
    # two clouds of different sizes, noisy un-normalised normals (the optimiser does not keep them unit length)
    sel_a = rng.permutation(pts.shape[0])[:1500]
    sel_b = rng.permutation(pts.shape[0])[:1100]
    clouds = []
    for sel, sc in ((sel_a, 1.0), (sel_b, 0.8)):
        p = (pts[sel] * sc + rng.normal(0, 0.004, (len(sel), 3))).astype(np.float32)
        n = (nrm[sel] * rng.uniform(0.7, 1.3, (len(sel), 1)) + rng.normal(0, 0.15, (len(sel), 3))).astype(np.float32)
        clouds.append((p, n))
    lengths = torch.tensor([c[0].shape[0] for c in clouds])
    maxp = int(lengths.max())
    P_pad = torch.zeros(2, maxp, 3)
    N_pad = torch.zeros(2, maxp, 3)
    for b, (p, n) in enumerate(clouds):
        P_pad[b, : len(p)] = torch.from_numpy(p)
        N_pad[b, : len(n)] = torch.from_numpy(n)
    vis = torch.zeros(2, maxp, dtype=torch.bool)
    inm = torch.zeros(2, maxp, dtype=torch.bool)
    for b in range(2):
        vis[b, : int(lengths[b])] = torch.from_numpy(rng.random(int(lengths[b])) < 0.6)
        inm[b, : int(lengths[b])] = torch.from_numpy(rng.random(int(lengths[b])) < 0.8)
    flt = types.SimpleNamespace(visibility=vis, inmask=inm)

    out = {"points_a": clouds[0][0], "normals_a": clouds[0][1], "points_b": clouds[1][0], "normals_b": clouds[1][1],
           "visibility": np.concatenate([vis[b, : int(lengths[b])].numpy() for b in range(2)]),
           "inmask": np.concatenate([inm[b, : int(lengths[b])].numpy() for b in range(2)])}

    def packed(x):
        return np.concatenate([x[b, : int(lengths[b])].detach().numpy() for b in range(2)])

    for knn_k, sigma, fscale in ((12, 0.75, 2.0), (33, 0.5, 1.0)):
        tag = "k%d" % knn_k
        # --- projection loss
        Pp = P_pad.clone().requires_grad_(True)
        pl = ref_losses.ProjectionLoss(reduction="none", knn_k=knn_k, filter_scale=fscale, sharpness_sigma=sigma)
        loss = pl(_Clouds(Pp, N_pad, lengths), rebuild_knn=True, points_filter=flt)  # (Ptotal,)
        g_up = torch.from_numpy(rng.normal(0, 1, loss.shape).astype(np.float32))
        (loss * g_up).sum().backward()
        out[tag + "_proj_loss"] = loss.detach().numpy()
        out[tag + "_proj_gup"] = g_up.numpy()
        out[tag + "_proj_grad"] = packed(Pp.grad)
        out[tag + "_proj_mean"] = np.float32(
            ref_losses.ProjectionLoss(reduction="mean", knn_k=knn_k, filter_scale=fscale, sharpness_sigma=sigma)(
                _Clouds(P_pad, N_pad, lengths), rebuild_knn=True, points_filter=flt).item())
        # mollified normals (shared first stage of both losses, losses.py:178-213)
        with torch.no_grad():
            phi = pl.get_phi(_Clouds(P_pad, N_pad, lengths))
            den = pl._denoise_normals(_Clouds(P_pad, N_pad, lengths), phi, flt)
        out[tag + "_mollified"] = packed(den.normals_padded())
        # --- repulsion loss.  get_spatial_w multiplies (N,P,K) by a (N,) factor (losses.py:252-258), which only
        # broadcasts for a batch of ONE cloud (what train_mvr.py optimises): run it cloud by cloud.
        losses_r, gups_r, grads_r = [], [], []
        for b in range(2):
            L = int(lengths[b])
            Pr = P_pad[b: b + 1, :L].clone().requires_grad_(True)
            fb = types.SimpleNamespace(visibility=vis[b: b + 1, :L], inmask=inm[b: b + 1, :L])
            rl = ref_losses.RepulsionLoss(reduction="none", knn_k=knn_k, filter_scale=fscale, sharpness_sigma=sigma)
            lossr = rl(_Clouds(Pr, N_pad[b: b + 1, :L], lengths[b: b + 1]), rebuild_knn=True, points_filter=fb)  # (L,3)
            g_upr = torch.from_numpy(rng.normal(0, 1, lossr.shape).astype(np.float32))
            (lossr * g_upr).sum().backward()
            losses_r.append(lossr.detach().numpy()); gups_r.append(g_upr.numpy()); grads_r.append(Pr.grad[0].numpy())
        out[tag + "_repel_loss"] = np.concatenate(losses_r)
        out[tag + "_repel_gup"] = np.concatenate(gups_r)
        out[tag + "_repel_grad"] = np.concatenate(grads_r)
        out[tag + "_params"] = np.array([knn_k, sigma, fscale], np.float32)
        print(tag, "proj mean", float(loss.mean()), "repel mean", float(out[tag + "_repel_loss"].mean()),
              "|grad proj|", float(Pp.grad.norm()), "|grad repel|", float(np.linalg.norm(out[tag + "_repel_grad"])))
    path = os.path.join(HERE, "ref_losses.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
