"""BASELINE configs[2] through the reference's OWN `train_mvr.py` (unmodified) on the HIP kernels -- shared by
`tests/test_gpu_reference_loop.py` and `tools/train_mvr_ref.py`.

The GPU box has no /root/reference; `make -C oracle ref_py` (run by `__graft_entry__.build()` where the reference exists)
packs the reference's Python tree -- train_mvr.py, config.py, common.py, configs/, DSS/**.py, the yoga6 scan -- into the
git-ignored build output `oracle/_ref/reference_py.tgz`, which travels with the snapshot like the compiled reference
`oracle/_ref/*.so`.  Here it is unpacked into a temporary directory and handed to `launcher.py --reference`.

Workload (train_mvr.py:158-161, configs/dss.yml:14-40, config.py:241-261): target = yoga6 scan x10 tangent-plane jitter
= 99,790 points rendered from 128 views drawn by the reference's CameraSampler at 512^2; model = a sphere of 99,790
points (config.py:177-183) with learnable positions and normals; batches of 8 views; dss.yml raster parameters; image loss
+ 0.01 x ProjectionLoss; Adam."""
import json
import os
import subprocess
import sys
import tarfile

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ARCHIVE = os.path.join(ROOT, "oracle", "_ref", "reference_py.tgz")
LAUNCHER = os.path.join(HERE, "launcher.py")

SIZE, VIEWS, BATCH, POINTS, JITTER = 512, 128, 8, 99790, 10


def reference_root(tmp):
    """the reference checkout to run: the staged archive (GPU box), else the live checkout (build container)"""
    if os.path.isfile(ARCHIVE):
        dst = os.path.join(tmp, "reference_py")
        if not os.path.isdir(dst):
            with tarfile.open(ARCHIVE) as t:
                t.extractall(dst)
        return dst
    if os.path.isfile("/root/reference/train_mvr.py"):
        return "/root/reference"
    raise RuntimeError("oracle/_ref/reference_py.tgz is missing: run `make -C oracle ref_py` (or __graft_entry__.build()) "
                       "on a machine that has the reference checkout; the archive then travels to the GPU box")


def write_configs(tmp, size=SIZE, points=POINTS, batch=BATCH, backup_every=0):
    """-> (class-level yml, c-level yml): dss.yml's parameters; the first selects the drop-in classes (INTEGRATION.md
    section 2), the second keeps the reference's OWN classes (configs/default.yaml) for `launcher --c-level`."""
    cfg = {
        "name": "dropin",
        "data": {"type": "MVR", "data_dir": os.path.join(tmp, "data"), "resolution": [size, size]},
        "model": {"type": "point", "model_kwargs": {"n_points_per_cloud": points, "learn_colors": False,
                                                    "learn_points": True, "learn_normals": True}},
        "renderer": {
            "is_neural_texture": False,
            "renderer_type": "dss_amd.renderer.SurfaceSplattingRenderer",
            "raster_type": "dss_amd.rasterizer.SurfaceSplatting",
            "compositor_type": "dss_amd.renderer.NormWeightedCompositor",
            "raster_params": {"Vrk_invariant": True, "Vrk_isotropic": False, "clip_pts_grad": 0.05,
                              "cutoff_threshold": 1.0, "depth_merging_threshold": 0.05, "image_size": size,
                              "points_per_pixel": 5, "radii_backward_scaler": 5},
        },
        # backup_every = k: train_mvr.py:192-196 saves model_<it>.pt at every k-th ITERATION (a snapshot by iteration count)
        "training": {"out_dir": os.path.join(tmp, "exp"), "backup_every": int(backup_every), "batch_size": batch, "checkpoint_every": 0,
                     "debug_every": 0, "visualize_every": 0, "validate_every": 0, "print_every": 1,
                     "lambda_dr_proj": 0.01, "lambda_dr_repel": 0.0, "lambda_dr_rgb": 1.0, "lambda_dr_silhouette": 1.0,
                     "n_workers": 0, "steps_dss_backward_radii": 200, "gamma_dss_backward_radii": 0.9,
                     "limit_dss_backward_radii": 2},
    }
    a = os.path.join(tmp, "dropin.yml")
    with open(a, "w") as f:
        yaml.safe_dump(cfg, f)
    cfg["name"] = "native"
    cfg["renderer"].update(renderer_type="DSS.core.renderer.SurfaceSplattingRenderer",
                           raster_type="DSS.core.rasterizer.SurfaceSplatting",
                           compositor_type="pytorch3d.renderer.NormWeightedCompositor")
    b = os.path.join(tmp, "native.yml")
    with open(b, "w") as f:
        yaml.safe_dump(cfg, f)
    return a, b


def run(args, timeout, prefix=()):
    env = dict(os.environ, OMP_NUM_THREADS="8")
    return subprocess.run(list(prefix) + [sys.executable, LAUNCHER] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, timeout=timeout, text=True)


def reached_time_limit(r):
    """train_mvr.py:219-228 leaves through exit(3) at its time limit -- after saving model.pt it joins
    `trainer._threads`, which only `Trainer.debug` creates (trainer.py:461): with debug_every 0 the unmodified script ends
    on that AttributeError instead."""
    return r.returncode == 3 or (r.returncode == 1 and "no attribute '_threads'" in r.stdout)


def losses(scalars):
    rows = [json.loads(l) for l in open(scalars)] if os.path.isfile(scalars) else []
    rows = [d for d in rows if d["tag"] == "train/loss"]
    return [d["value"] for d in rows], [d["step"] for d in rows], [d.get("t", 0.0) for d in rows]


def ms_per_iteration(times, steps):
    """median wall-clock distance of consecutive logged iterations (the legs' start-up and the resume gaps drop out)"""
    d = sorted(1e3 * (times[i + 1] - times[i]) for i in range(len(times) - 1) if steps[i + 1] == steps[i] + 1)
    return d[len(d) // 2] if d else float("nan")
