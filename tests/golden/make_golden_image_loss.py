"""Generates tests/golden/ref_image_loss.npz by calling the REFERENCE's own `Trainer.calc_dr_loss`
(/root/reference/DSS/training/trainer.py:332-372: masked L1 on RGB through `L1Loss`, silhouette L1 + 0.01 IoU through
`IouLoss`, losses.py:127-135, :498-513, reductions of `BaseLoss` :42-61) with autograd providing the gradients with
respect to the predicted image and mask.

The Trainer class itself cannot be constructed here (tensorboard, datasets, ...): the method only reads
`lambda_dr_silhouette`, `lambda_dr_rgb`, `l1_loss`, `iou_loss` from `self`, so it is called unbound on a namespace
holding the reference's loss objects built exactly as `Trainer.__init__` builds them (trainer.py:138-141).  Stubs as
in make_golden_setup.py, plus an inert `torch.utils.tensorboard`.

    python tests/golden/make_golden_image_loss.py
"""
import importlib
import os
import sys

sys.dont_write_bytecode = True  # the reference checkout is read-only: importing it must not leave __pycache__ there
import types

import numpy as np
import torch

import make_golden_setup as base
import pytorch3d.ops.knn as ops3d_knn
import pytorch3d.renderer as p3r
import pytorch3d.renderer.lighting as p3r_lighting

HERE = os.path.dirname(os.path.abspath(__file__))
ops3d_knn._KNN = base._KNN
p3r.lighting = p3r_lighting
sys.modules["torch.utils.tensorboard"] = base._StubModule("torch.utils.tensorboard")
ref_trainer = importlib.import_module("DSS.training.trainer")  # the UNMODIFIED reference module
ref_losses = importlib.import_module("DSS.training.losses")


def run_case(out, tag, N, H, W, lam_rgb, lam_sil, rng, empty_overlap=False):
    img = rng.random((N, H, W, 3)).astype(np.float32)
    pred = rng.random((N, H, W, 3)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    mask = np.stack([((yy - H * (0.4 + 0.1 * n)) ** 2 + (xx - W * 0.5) ** 2 < (0.3 * min(H, W)) ** 2) for n in range(N)])
    mask_pred = np.stack([((yy - H * 0.5) ** 2 + (xx - W * (0.45 + 0.05 * n)) ** 2 < (0.33 * min(H, W)) ** 2)
                          for n in range(N)])
    if empty_overlap:
        mask_pred = ~mask & (rng.random((N, H, W)) < 0.3)
    if N > 2:
        mask[-1] = False                      # an image without any target silhouette
    pred[0, : H // 4] = img[0, : H // 4]      # exact ties: |x|' = 0 there
    mask, mask_pred = mask.astype(np.float32), mask_pred.astype(np.float32)

    fake = types.SimpleNamespace(lambda_dr_silhouette=lam_sil, lambda_dr_rgb=lam_rgb,
                                 l1_loss=ref_losses.L1Loss(reduction="mean"),
                                 iou_loss=ref_losses.IouLoss(reduction="mean", channel_dim=None))
    t_pred = torch.from_numpy(pred).requires_grad_(True)
    t_mp = torch.from_numpy(mask_pred).requires_grad_(True)
    loss = {"loss": 0}
    ref_trainer.Trainer.calc_dr_loss(fake, torch.from_numpy(img), t_pred, torch.from_numpy(mask), t_mp,
                                     reduction_method="mean", loss=loss)
    loss["loss"].backward()
    out[tag + "_img"], out[tag + "_pred"], out[tag + "_mask"], out[tag + "_mask_pred"] = img, pred, mask, mask_pred
    out[tag + "_lambdas"] = np.array([lam_rgb, lam_sil], np.float32)
    val = {k: (v.item() if torch.is_tensor(v) else float(v)) for k, v in loss.items()}
    out[tag + "_loss"] = np.float32(val["loss"])
    out[tag + "_loss_rgb"] = np.float32(val["loss_dr_rgb"])
    out[tag + "_loss_sil"] = np.float32(val["loss_dr_silhouette"])
    out[tag + "_grad_pred"] = t_pred.grad.numpy() if t_pred.grad is not None else np.zeros_like(pred)
    out[tag + "_grad_mask_pred"] = t_mp.grad.numpy()
    print(tag, val)


def main():
    rng = np.random.default_rng(3)
    out = {}
    run_case(out, "a", 2, 64, 64, 1.0, 1.0, rng)          # configs/dss.yml:32-33
    run_case(out, "b", 3, 48, 40, 0.7, 2.0, rng)
    run_case(out, "c", 2, 32, 32, 1.0, 1.0, rng, empty_overlap=True)   # no pixel inside both masks: rgb term skipped
    path = os.path.join(HERE, "ref_image_loss.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
