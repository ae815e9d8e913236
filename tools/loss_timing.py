"""Time the regulariser chain of one training iteration (trainer.py:319-326): self kNN (K = 12), normal mollification,
projection loss forward + backward, repulsion loss forward + backward, on one cloud of P points.

    python tools/loss_timing.py [P]
"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, __import__("os").path.join(sys.path[0], "tests"))
import scenes  # noqa: E402
from dss_amd import ops  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 99790
    pts, nrm, _ = scenes.synthetic_cloud(P, seed=0)
    dev = "cuda:0"
    X, Nn = torch.from_numpy(pts).to(dev), torch.from_numpy(nrm).to(dev)
    F = torch.zeros(1, dtype=torch.int64, device=dev)
    L = torch.full((1,), P, dtype=torch.int64, device=dev)
    vis = torch.rand(P, device=dev) < 0.5
    g1 = torch.full((P,), 1.0 / P, device=dev)
    g3 = torch.full((P, 3), 1.0 / (3 * P), device=dev)
    dists, idx = ops.knn_points(X, F, L, 12)
    moll = ops.mollify_normals(Nn, dists, idx, vis, F, L)
    out = {"points": P, "knn_k": 12}
    out["knn_us"] = timed(lambda: ops.knn_points(X, F, L, 12))
    out["mollify_us"] = timed(lambda: ops.mollify_normals(Nn, dists, idx, vis, F, L))
    out["projection_fwd_us"] = timed(lambda: ops.projection_loss(X, moll, dists, idx, vis, F, L, 0.75))
    out["projection_bwd_us"] = timed(lambda: ops.projection_loss(X, moll, dists, idx, vis, F, L, 0.75, grad_loss=g1,
                                                                 want_loss=False, want_grad=True))
    out["repulsion_fwd_us"] = timed(lambda: ops.repulsion_loss(X, moll, idx, F, L, 0.75, 2.0))
    out["repulsion_bwd_us"] = timed(lambda: ops.repulsion_loss(X, moll, idx, F, L, 0.75, 2.0, grad_loss=g3,
                                                               want_loss=False, want_grad=True))
    # bytes one projection call must move: the (P,K) lists (12 B/entry: d2 + int64 id) + own point/normal + outputs;
    # the neighbour gathers (24 B each) are L2 hits for clouds of this size
    alg = P * (12 * 12 + 24 + 4)
    out["projection_fwd_GBps_algorithmic"] = alg / out["projection_fwd_us"] / 1e3
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
