"""SURVEY §8 row (g), BASELINE configs[2]: the reference's OWN `train_mvr.py` -- unmodified, from the staged archive
`oracle/_ref/reference_py.tgz` -- on the HIP kernels of one MI355X: 99,790-point target (yoga6 x10) seen from 128
CameraSampler views at 512^2, a 99,790-point model, batches of 8, dss.yml raster parameters, >= 200 iterations.

Two legs (tests/ref_loop/launcher.py):
  class level  the YAML names the drop-in classes; everything else is the reference's code;
  C level      the YAML keeps the reference's own rasterizer / renderer classes and only `DSS._C` is `dss_amd.ops`.
No oracle, no `--no-cuda`: `dss_amd.ops` is the real library (the launcher installs the oracle double only without a GPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_loop"))
import cfg3  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500)
def test_reference_train_mvr_unmodified_on_the_hip_kernels_at_configs2(tmp_path):
    tmp = str(tmp_path)
    ref = cfg3.reference_root(tmp)
    cfg_cls, cfg_c = cfg3.write_configs(tmp)
    sc_cls, sc_c = os.path.join(tmp, "scalars.jsonl"), os.path.join(tmp, "scalars_native.jsonl")
    common = ["--reference", ref]
    r = cfg3.run(common + ["--config", cfg_cls, "--make-dataset", os.path.join(tmp, "data"), "--views", str(cfg3.VIEWS),
                           "--jitter", str(cfg3.JITTER), "--camera-sampler"], 600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "oracle_ops" not in r.stdout
    assert len(os.listdir(os.path.join(tmp, "data", "image"))) == cfg3.VIEWS

    # ---- class-level leg: >= 200 iterations (train_mvr.py stops on wall clock and resumes from its own model.pt)
    loss, legs = [], 0
    while len(loss) < 200 and legs < 5:
        legs += 1
        r = cfg3.run(common + ["--config", cfg_cls, "--scalars", sc_cls, "--exit-after", "40"], 600)
        assert cfg3.reached_time_limit(r), r.stdout[-4000:]
        loss, steps, times = cfg3.losses(sc_cls)
        assert steps == sorted(steps) and len(set(steps)) == len(steps)
    assert len(loss) >= 200, (len(loss), legs, r.stdout[-2000:])
    assert os.path.isfile(os.path.join(tmp, "exp", "dropin", "model.pt"))
    n = len(loss)
    deciles = [sum(loss[i * n // 10:(i + 1) * n // 10]) / ((i + 1) * n // 10 - i * n // 10) for i in range(10)]
    ms_cls = cfg3.ms_per_iteration(times, steps)
    print("class-level leg: %d iterations, %.1f ms/iteration, loss deciles %s" % (n, ms_cls, ["%.4f" % d for d in deciles]))
    assert all(l == l and l < 1e3 for l in loss)
    # From the sphere of config.py:177-183 the loss the Trainer logs RISES at this scale (0.36 -> ~1.0 by iteration 300, then
    # a slow decline; measured identically through the reference's own rasterizer classes in the C-level leg below): with
    # 99,790 points the splats are ~1.5 px, only points within a splat radius of the target's outline get a coherent
    # occupancy gradient, and the script's hard-coded Adam(lr 0.01) (train_mvr.py:84-94) turns the residual of everything
    # else into 0.01-sized steps, so the silhouette inflates (the 1,500-point CPU leg of test_reference_loop_cpu.py shows
    # the same inflation for its first ~150 iterations).  That is the reference's optimisation, not the renderer: what the
    # test requires of this leg is that the loop runs, stays finite and is reproduced by the C-level leg.

    # ---- C-level leg: the reference's own classes on DSS._C = dss_amd.ops; the first epoch (128 views / 8 = 16
    # iterations) draws the same batches (afterwards rasterizer.py:334's torch.rand_like shifts the random stream)
    native, legs = [], 0
    while len(native) < 16 and legs < 4:
        legs += 1
        r = cfg3.run(common + ["--config", cfg_c, "--scalars", sc_c, "--c-level", "--exit-after", "40"], 900)
        assert cfg3.reached_time_limit(r), r.stdout[-4000:]
        native, nsteps, ntimes = cfg3.losses(sc_c)
    assert len(native) >= 16, (len(native), r.stdout[-2000:])
    rel = [abs(a - b) / abs(b) for a, b in zip(native[:16], loss[:16])]
    print("C-level leg: %d iterations, %.1f ms/iteration, first-epoch loss rel. differences max %.2e"
          % (len(native), cfg3.ms_per_iteration(ntimes, nsteps), max(rel)))
    print("first epoch, class level:", ["%.6f" % v for v in loss[:16]])
    print("first epoch, C level    :", ["%.6f" % v for v in native[:16]])
    print("first epoch, rel. differences:", ["%.1e" % v for v in rel])
    # identical batches, identical kernels underneath; the reference's own rasterizer classes evaluate h, the Jacobian and
    # the blend in another operation order, and Adam turns that fp32 round-off into a slowly growing difference of the
    # trajectories: 1e-7 at the first iteration, 3e-5 by the fourth, 3e-3 by the sixteenth (measured)
    assert max(rel[:4]) <= 1e-4 and max(rel) <= 2e-2, rel

    # ---- the same unmodified script RESUMING (train_mvr.py:98-103, `resume_from: model.pt`) from a noisy copy of the
    # target cloud (positions + N(0, 0.01), normals + N(0, 0.5) renormalised): the regime in which the surrogate gradient
    # has signal everywhere.  Here the loss the reference's Trainer logs must fall.
    import yaml
    c = yaml.safe_load(open(cfg_cls))
    c["name"] = "resume"
    cfg_res, sc_res = os.path.join(tmp, "resume.yml"), os.path.join(tmp, "scalars_resume.jsonl")
    yaml.safe_dump(c, open(cfg_res, "w"))
    r = cfg3.run(common + ["--config", cfg_res, "--make-checkpoint", os.path.join(tmp, "exp", "resume", "model.pt"),
                           "--data-dir", os.path.join(tmp, "data"), "--noise", "0.01"], 300)
    assert r.returncode == 0, r.stdout[-3000:]
    r = cfg3.run(common + ["--config", cfg_res, "--scalars", sc_res, "--exit-after", "25"], 600)
    assert cfg3.reached_time_limit(r), r.stdout[-4000:]
    assert "Loading checkpoint from local file" in r.stdout
    res, rsteps, rtimes = cfg3.losses(sc_res)
    n = len(res)
    assert n >= 200, n
    deciles = [sum(res[i * n // 10:(i + 1) * n // 10]) / ((i + 1) * n // 10 - i * n // 10) for i in range(10)]
    print("resumed leg: %d iterations, %.1f ms/iteration, loss deciles %s"
          % (n, cfg3.ms_per_iteration(rtimes, rsteps), ["%.4f" % d for d in deciles]))
    assert deciles[-1] < 0.93 * deciles[0] and min(deciles[5:]) < 0.9 * deciles[0], deciles   # measured: 0.253 -> 0.21


@pytest.mark.timeout(1200)
def test_reference_train_mvr_converges_from_the_sphere_at_the_size_of_dss_yml(tmp_path):
    """the configs[2]-sized model above is 20x larger than what the reference's own `configs/dss.yml:8`
    trains (`n_points_per_cloud: 5000`), and from the sphere its loss rises.  Here the SAME unmodified script and dataset
    (99,790-point yoga6 target, 128 CameraSampler views at 512^2, batches of 8, dss.yml raster parameters and weights) with
    dss.yml's own model size: the loss the reference's Trainer logs must FALL (last decile < 0.8 x first decile) and the
    model must move ONTO the target surface: the median model -> target distance and the target -> model distance (coverage)
    fall between the snapshot after FIVE iterations (`backup_every: 5` -> model_5.pt, train_mvr.py:192-196: by iteration count,
    not by wall clock) and the end.  The symmetric chamfer distance (`Trainer.evaluate_3d`, trainer.py:144) does NOT fall: its
    model -> target MEAN is dominated by the ~30 % of the points that the reference's optimisation (Adam(lr 0.01), no pruning:
    point_modeling.py:131-132 is commented out) carries out of the view volume.  That this belongs to the reference and not to
    the HIP gradient is MEASURED on the CPU by `tools/convergence_crosscheck_cpu.py` (profiles/r5_a_convergence_crosscheck_cpu.json):
    the reference's own rasterizer / renderer classes on the oracle double lose the same fraction of the sphere."""
    import json
    import numpy as np
    import torch
    from scipy.spatial import cKDTree
    tmp = str(tmp_path)
    ref = cfg3.reference_root(tmp)
    cfg_cls, _ = cfg3.write_configs(tmp, points=5000, backup_every=5)
    sc = os.path.join(tmp, "scalars_5000.jsonl")
    common = ["--reference", ref]
    r = cfg3.run(common + ["--config", cfg_cls, "--make-dataset", os.path.join(tmp, "data"), "--views", str(cfg3.VIEWS),
                           "--jitter", str(cfg3.JITTER), "--camera-sampler"], 600)
    assert r.returncode == 0, r.stdout[-3000:]
    target = np.load(os.path.join(tmp, "data", "data_dict.npz"), allow_pickle=True)["points"].astype(np.float64)
    tree_t = cKDTree(target)

    def chamfer(model_pt):
        """squared-distance statistics between the model's points and the target cloud: the symmetric chamfer distance
        (mean over both directions), its two halves, and the median model -> target distance"""
        pts = torch.load(model_pt, map_location="cpu")["model"]["points"].reshape(-1, 3).double().numpy()
        d_mt, _ = tree_t.query(pts)
        d_tm, _ = cKDTree(pts).query(target)
        return {"chamfer": float((d_mt ** 2).mean() + (d_tm ** 2).mean()), "target_to_model": float((d_tm ** 2).mean()),
                "model_to_target": float((d_mt ** 2).mean()), "model_to_target_median": float(np.median(d_mt)),
                "model_points_farther_than_0.2": float((d_mt > 0.2).mean())}

    model_pt = os.path.join(tmp, "exp", "dropin", "model.pt")
    # the early snapshot is taken by ITERATION COUNT: the first leg runs with `backup_every: 5` (train_mvr.py saves
    # model_5.pt, model_10.pt, ... at those iterations) for a few seconds; the later legs run without backups.  (Round 4 took
    # the snapshot after one second of wall clock: 50-160 iterations depending on the box)
    r = cfg3.run(common + ["--config", cfg_cls, "--scalars", sc, "--exit-after", "3"], 600)
    assert cfg3.reached_time_limit(r), r.stdout[-4000:]
    early_pt = os.path.join(tmp, "exp", "dropin", "model_5.pt")
    assert os.path.isfile(early_pt), os.listdir(os.path.join(tmp, "exp", "dropin"))
    cd_early = chamfer(early_pt)
    cfg_cls, _ = cfg3.write_configs(tmp, points=5000, backup_every=0)
    loss, legs = [], 0
    while len(loss) < 1200 and legs < 4:   # resumes from its own model.pt (train_mvr.py:98-103)
        legs += 1
        r = cfg3.run(common + ["--config", cfg_cls, "--scalars", sc, "--exit-after", "45"], 600)
        assert cfg3.reached_time_limit(r), r.stdout[-4000:]
        loss, steps, times = cfg3.losses(sc)
    cd_late = chamfer(model_pt)
    n = len(loss)
    assert n >= 600, (n, legs)
    deciles = [sum(loss[i * n // 10:(i + 1) * n // 10]) / ((i + 1) * n // 10 - i * n // 10) for i in range(10)]
    rec = {"points_per_cloud": 5000, "iterations": n, "ms_per_iteration": cfg3.ms_per_iteration(times, steps),
           "loss_deciles": deciles, "chamfer_after_5_iterations": cd_early, "chamfer_at_end": cd_late}
    print("dss.yml-sized model from the sphere:", json.dumps(rec))
    print("distances after the first iterations:", cd_early)
    print("distances at the end               :", cd_late)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "train_mvr_ref_5000.json"), "w"), indent=1)
    except OSError:
        pass
    assert all(l == l and l < 1e3 for l in loss)
    assert deciles[-1] < 0.8 * deciles[0], deciles
    # The model moves onto the target: the typical model point ends up on the target's surface (median model -> target
    # distance: 0.30 -> 0.034 measured) and the target is covered better (target -> model: 7.5e-2 -> 0.69e-3).  The MEAN
    # model -> target distance -- and with it the symmetric chamfer distance -- RISES (0.13 -> 0.77): the reference neither
    # prunes nor bounds its points (point_modeling.py:131-132 is commented out), and its hard-coded Adam(lr 0.01)
    # (train_mvr.py:84-94) carries the ~30 % of the sphere's points whose silhouette gradient keeps its sign out of the
    # view volume, 0.01 per iteration.  That is the reference's optimisation (the C-level leg of the test above reproduces
    # the class-level trajectory); both halves are recorded.
    assert cd_late["model_to_target_median"] < 0.5 * cd_early["model_to_target_median"], (cd_early, cd_late)
    # (coverage of the target: 0.69 ... 0.72e-3 at the end in every run; after five iterations the model is practically the
    # initial sphere: 7.5e-2)
    assert cd_late["target_to_model"] < 0.1 * cd_early["target_to_model"] and cd_late["target_to_model"] < 0.9e-3, (cd_early, cd_late)
    assert cd_late["model_points_farther_than_0.2"] < cd_early["model_points_farther_than_0.2"], (cd_early, cd_late)


@pytest.mark.timeout(900)
def test_reference_train_mvr_unmodified_under_torchrun_with_the_row_sharded_renderer(tmp_path):
    """VERDICT r5 item 1 ("a training loop cannot select it from YAML today"): the reference's unmodified `train_mvr.py`
    started by `torch.distributed.run` with TWO ranks, its YAML naming `dss_amd.renderer.RowShardedSurfaceSplattingRenderer`
    (config.py:241-261 can only pass a class path).  Every rank runs the same script on the same batches and renders its own
    image rows; the class joins the process group in its constructor (gloo here: both ranks share the one GPU of the test box;
    RCCL on a multi-GPU node).  The loss the Trainer logs must be THE SAME on both ranks in every iteration (same full image,
    same all-reduced gradients -> same parameters) and must follow the single-GPU run of the same YAML with the plain
    renderer (identical batches; fp32 reduction order differs)."""
    import yaml
    tmp = str(tmp_path)
    ref = cfg3.reference_root(tmp)
    cfg_cls, _ = cfg3.write_configs(tmp, size=256, points=5000, batch=4)
    common = ["--reference", ref]
    r = cfg3.run(common + ["--config", cfg_cls, "--make-dataset", os.path.join(tmp, "data"), "--views", "16", "--jitter", "2",
                           "--camera-sampler"], 600)
    assert r.returncode == 0, r.stdout[-3000:]
    # single GPU, plain renderer
    sc1 = os.path.join(tmp, "scalars_single.jsonl")
    r = cfg3.run(common + ["--config", cfg_cls, "--scalars", sc1, "--exit-after", "12"], 600)
    assert cfg3.reached_time_limit(r), r.stdout[-4000:]
    single, _, _ = cfg3.losses(sc1)
    assert len(single) >= 12, len(single)
    # two ranks, the row-sharded renderer selected by the YAML
    c = yaml.safe_load(open(cfg_cls))
    c["name"] = "sharded"
    c["renderer"]["renderer_type"] = "dss_amd.renderer.RowShardedSurfaceSplattingRenderer"
    cfg_sh, sc2 = os.path.join(tmp, "sharded.yml"), os.path.join(tmp, "scalars_sharded.jsonl")
    yaml.safe_dump(c, open(cfg_sh, "w"))
    os.makedirs(os.path.join(tmp, "exp", "sharded"), exist_ok=True)   # (train_mvr.py:54-55 is not written for two processes)
    env_keep = dict(os.environ)
    os.environ["DSS_AMD_DIST_BACKEND"] = "gloo"
    try:
        r = cfg3.run(common + ["--config", cfg_sh, "--scalars", sc2, "--exit-after", "12"], 800,
                     prefix=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                             "127.0.0.1", "--master-port", "29741", "--no-python"])
    finally:
        os.environ.clear()
        os.environ.update(env_keep)
    out = r.stdout
    assert "Time limit reached" in out or cfg3.reached_time_limit(r), out[-6000:]
    l0, s0, _ = cfg3.losses(sc2 + ".rank0")
    l1, s1, _ = cfg3.losses(sc2 + ".rank1")
    n = min(len(l0), len(l1))
    assert n >= 8, (len(l0), len(l1), out[-3000:])
    # the ranks hold the same parameters after every step: the logged loss is bit-identical
    assert l0[:n] == l1[:n], [(a, b) for a, b in zip(l0[:n], l1[:n]) if a != b][:4]
    m = min(n, len(single), 8)
    rel = [abs(a - b) / abs(b) for a, b in zip(l0[:m], single[:m])]
    print("row-sharded (2 ranks) vs single GPU, first %d iterations, rel. loss differences: %s" % (m, ["%.1e" % v for v in rel]))
    assert max(rel[:3]) <= 1e-4 and max(rel) <= 2e-2, rel
