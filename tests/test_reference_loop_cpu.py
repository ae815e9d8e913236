"""the reference's own `train_mvr.py`, `config.py`, `DSS.training.trainer`, `DSS.models.*`,
`DSS.core.cloud` and `DSS.utils.dataset` -- imported UNMODIFIED from /root/reference -- run on top of the drop-in
classes, with only the YAML class paths of INTEGRATION.md §2 changed, through the `compat/pytorch3d` namespace package.

The build container has no GPU and the GPU box has no /root/reference, so the one place where both the reference loop and
this repository exist is here, on the CPU: `tests/ref_loop/launcher.py` answers `dss_amd.ops` with the oracle (a test
double at the C-ABI seam; the product itself still refuses to run without the HIP library) and the test checks that the
reference's loop trains: iterations complete, the loss it logs decreases.  With a GPU *and* a reference checkout the same
launcher runs the real kernels (`python tests/ref_loop/launcher.py --config ... --scalars ...` without `--no-cuda`)."""
import json
import os
import subprocess
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LAUNCHER = os.path.join(ROOT, "tests", "ref_loop", "launcher.py")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train_mvr.py")),
                                reason="needs the reference checkout (absent on the GPU box)")


def _config(tmp, size):
    cfg = {
        "name": "dropin",
        "data": {"type": "MVR", "data_dir": os.path.join(tmp, "data"), "resolution": [size, size]},
        "model": {"type": "point", "model_kwargs": {"n_points_per_cloud": 1500, "learn_colors": False,
                                                    "learn_points": True, "learn_normals": True}},
        "renderer": {   # the only lines that differ from configs/dss.yml
            "is_neural_texture": False,
            "renderer_type": "dss_amd.renderer.SurfaceSplattingRenderer",
            "raster_type": "dss_amd.rasterizer.SurfaceSplatting",
            "compositor_type": "dss_amd.renderer.NormWeightedCompositor",
            "raster_params": {"Vrk_invariant": True, "Vrk_isotropic": False, "clip_pts_grad": 0.05,
                              "cutoff_threshold": 1.0, "depth_merging_threshold": 0.05, "image_size": size,
                              "points_per_pixel": 5, "radii_backward_scaler": 5},
        },
        "training": {"out_dir": os.path.join(tmp, "exp"), "backup_every": 0, "batch_size": 4, "checkpoint_every": 0,
                     "debug_every": 0, "visualize_every": 0, "validate_every": 0, "print_every": 1,
                     "lambda_dr_proj": 0.01, "lambda_dr_repel": 0.0, "lambda_dr_rgb": 1.0, "lambda_dr_silhouette": 1.0,
                     "n_workers": 0, "steps_dss_backward_radii": 200, "gamma_dss_backward_radii": 0.9,
                     "limit_dss_backward_radii": 2},
    }
    path = os.path.join(tmp, "dropin.yml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def _run(args, timeout):
    env = dict(os.environ, OMP_NUM_THREADS="8")
    return subprocess.run([sys.executable, LAUNCHER] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                          timeout=timeout, text=True)


@pytest.mark.timeout(1800)
def test_reference_train_mvr_runs_unmodified_on_the_drop_in_and_its_loss_decreases(tmp_path):
    tmp = str(tmp_path)
    cfg = _config(tmp, 192)   # (at 64^2 the splats and the backward radius rs = 5 x median radius are a quarter of the object:
    # the surrogate gradient then inflates the silhouette for hundreds of iterations; 192^2 is in the regime of the 512^2 configs)
    scalars = os.path.join(tmp, "scalars.jsonl")
    r = _run(["--config", cfg, "--scalars", scalars, "--no-cuda", "--make-dataset", os.path.join(tmp, "data"),
              "--views", "16", "--target-points", "3000"], 300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert len(os.listdir(os.path.join(tmp, "data", "image"))) == 16
    # train_mvr.py stops on a wall-clock limit, not an iteration count: run legs of 60 s (it resumes from its own model.pt,
    # train_mvr.py:98-103) until 450 iterations are in, so that a slow or busy machine does not decide the outcome
    loss, legs = [], 0
    while len(loss) < 450 and legs < 8:
        legs += 1
        r = _run(["--config", cfg, "--scalars", scalars, "--no-cuda", "--exit-after", "60"], 400)
        # train_mvr.py:219-228 leaves through exit(3) when its time limit is reached -- after saving model.pt it joins
        # `trainer._threads`, an attribute only `Trainer.debug` creates (trainer.py:461): without a debug visualisation
        # (debug_every: 0 here, it needs plotly / trimesh) the unmodified script ends on that AttributeError instead
        reached_time_limit = r.returncode == 3 or (r.returncode == 1 and "no attribute '_threads'" in r.stdout)
        assert reached_time_limit, r.stdout[-4000:]
        assert os.path.isfile(os.path.join(tmp, "exp", "dropin", "model.pt"))  # the reference's CheckpointIO wrote it
        loss = [json.loads(l) for l in open(scalars)]
        steps = [d["step"] for d in loss if d["tag"] == "train/loss"]
        loss = [d["value"] for d in loss if d["tag"] == "train/loss"]
        assert steps == sorted(steps) and len(set(steps)) == len(steps), "a resumed leg continues the iteration count"
    assert len(loss) >= 300, (len(loss), legs, r.stdout[-2000:])
    # the surrogate gradient first inflates the sphere's silhouette for ~150 iterations, then the loss falls: 0.46 -> 0.34
    # after ~450 iterations, 0.21 after 1200 (6-10 iterations per second here, depending on the load of the machine)
    deciles = [sum(loss[i * len(loss) // 10:(i + 1) * len(loss) // 10]) / max(1, (i + 1) * len(loss) // 10 - i * len(loss) // 10)
               for i in range(10)]
    assert deciles[-1] < deciles[0] and deciles[-1] <= 0.92 * max(deciles), deciles
    assert os.path.isfile(os.path.join(tmp, "exp", "dropin", "model.pt"))  # the reference's CheckpointIO wrote it

    # The same script one seam lower: the YAML keeps the reference's OWN class paths (configs/default.yaml) and only
    # `DSS._C` is replaced by `dss_amd.ops` (launcher --c-level; FRNN / prefix_sum stand-ins for the reference's Python
    # grid build).  The first epoch (16 views / batches of 4) draws the same batches in both runs -- afterwards the
    # reference's `torch.rand_like` in rasterizer.py:334 shifts the random stream -- and must log the same losses.
    import yaml as _yaml
    c = _yaml.safe_load(open(cfg))
    c["name"] = "native"
    c["renderer"].update(renderer_type="DSS.core.renderer.SurfaceSplattingRenderer",
                         raster_type="DSS.core.rasterizer.SurfaceSplatting",
                         compositor_type="pytorch3d.renderer.NormWeightedCompositor")
    cfg_native, scalars_native = os.path.join(tmp, "native.yml"), os.path.join(tmp, "scalars_native.jsonl")
    _yaml.safe_dump(c, open(cfg_native, "w"))
    r = _run(["--config", cfg_native, "--scalars", scalars_native, "--no-cuda", "--c-level", "--exit-after", "12"], 300)
    assert r.returncode == 3 or (r.returncode == 1 and "no attribute '_threads'" in r.stdout), r.stdout[-4000:]
    native = [json.loads(l) for l in open(scalars_native)]
    native = [d["value"] for d in native if d["tag"] == "train/loss"]
    assert len(native) >= 12, len(native)
    for a, b in zip(native[:4], loss[:4]):
        assert abs(a - b) <= 2e-3 * abs(b), (native[:4], loss[:4])   # observed: 5e-5


def test_reference_rasterizer_classes_run_on_the_c_level_drop_in():
    """One seam lower: the reference's OWN `DSS.core.rasterizer.SurfaceSplatting` / `EllipticalRasterizer` /
    `DSS.core.renderer.SurfaceSplattingRenderer` (unmodified) with `DSS._C = dss_amd.ops` -- the same-name mirrors of
    the compiled extension (ext.cpp:5-18) -- against the drop-in classes: same fragments, same
    image, same gradients.  The reference's backward (rasterizer.py:787-977) first builds an FRNN grid with lxxue/FRNN and
    lxxue/prefix_sum, absent here: the launcher supplies stand-ins of their published behaviour; the mirror of
    `_splat_points_occ_fast_cuda_backward` ignores the grid arguments anyway."""
    r = _run(["--check-c-seam", "--no-cuda"], 300)
    line = [l for l in r.stdout.splitlines() if l.startswith("C_SEAM ")]
    assert r.returncode == 0 and line, r.stdout[-3000:]
    d = json.loads(line[-1][len("C_SEAM "):])
    assert d["fragments"] > 10000 and 0.1 < d["coverage"] < 0.5
    assert d["idx_equal_fraction"] >= 0.9995 and d["occupancy_equal_fraction"] >= 0.9995   # observed: 1.0 and 1.0
    assert d["qvalue_max_abs_diff_on_equal"] <= 1e-3 and d["zbuf_max_abs_diff_on_equal"] <= 1e-5
    assert d["scaler_rel_max_diff_on_equal"] <= 1e-3 and d["image_max_abs_diff"] <= 1e-4 and d["same_cloud_returned"]
    # backward through the reference's own autograd.Function, clip hook, projection and compositor
    assert d["grad_points_finite"] and d["grad_points_norm"] > 0.1
    assert d["grad_points_rel_l2"] <= 1e-3 and d["grad_colors_rel_l2"] <= 1e-3        # observed: 1.2e-5 and 4.5e-7
