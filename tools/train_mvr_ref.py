#!/usr/bin/env python3
"""Developer tool (GPU box): ms per iteration and a rocprofv3 kernel summary of the reference's OWN train_mvr.py on the
HIP kernels at BASELINE configs[2] (same workload as tests/test_gpu_reference_loop.py, see tests/ref_loop/cfg3.py).

    python tools/train_mvr_ref.py OUT_DIR [seconds]        # writes OUT_DIR/train_mvr_ref.json (+ kernel stats CSV)
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_loop"))
import cfg3  # noqa: E402


def main():
    out = os.path.abspath(sys.argv[1])
    seconds = sys.argv[2] if len(sys.argv) > 2 else "40"
    os.makedirs(out, exist_ok=True)
    tmp = "/tmp/train_mvr_ref"
    os.makedirs(tmp, exist_ok=True)
    ref = cfg3.reference_root(tmp)
    cfg_cls, cfg_c = cfg3.write_configs(tmp)
    if not os.path.isdir(os.path.join(tmp, "data", "image")):
        r = cfg3.run(["--reference", ref, "--config", cfg_cls, "--make-dataset", os.path.join(tmp, "data"), "--views",
                      str(cfg3.VIEWS), "--jitter", str(cfg3.JITTER), "--camera-sampler"], 900)
        assert r.returncode == 0, r.stdout[-3000:]
    res = {"workload": "reference train_mvr.py (unmodified) on the HIP kernels: %d-point model, %d views, %d^2, batch %d"
                       % (cfg3.POINTS, cfg3.VIEWS, cfg3.SIZE, cfg3.BATCH)}
    legs = (("class_level", cfg_cls, []), ("c_level", cfg_c, ["--c-level"]))
    if os.environ.get("REF_LEGS") == "class":
        legs = legs[:1]
    for name, cfg, extra in legs:
        sc = os.path.join(tmp, "scalars_%s.jsonl" % name)
        if os.path.exists(sc):
            os.remove(sc)
        shutil.rmtree(os.path.join(tmp, "exp"), ignore_errors=True)
        r = cfg3.run(["--reference", ref, "--config", cfg, "--scalars", sc, "--exit-after", seconds] + extra, 900)
        assert cfg3.reached_time_limit(r), r.stdout[-3000:]
        loss, steps, times = cfg3.losses(sc)
        ms = cfg3.ms_per_iteration(times, steps)
        shutil.copy(sc, os.path.join(out, "scalars_%s.jsonl" % name))
        for sub in ("dropin", "native"):
            m = os.path.join(tmp, "exp", sub, "model.pt")
            if os.path.isfile(m):
                shutil.copy(m, os.path.join(out, "model_%s.pt" % name))
        res[name] = {"iterations": len(loss), "ms_per_iteration": ms, "loss_first": loss[0], "loss_last": loss[-1],
                     "Msplats_per_s": cfg3.BATCH * cfg3.POINTS / ms * 1e-3}
    # resumed from a noisy copy of the target (see tests/test_gpu_reference_loop.py)
    import yaml
    c = yaml.safe_load(open(cfg_cls))
    c["name"] = "resume"
    cfg_res, sc = os.path.join(tmp, "resume.yml"), os.path.join(tmp, "scalars_resume.jsonl")
    yaml.safe_dump(c, open(cfg_res, "w"))
    if os.path.exists(sc):
        os.remove(sc)
    r = cfg3.run(["--reference", ref, "--config", cfg_res, "--make-checkpoint", os.path.join(tmp, "exp", "resume", "model.pt"),
                  "--data-dir", os.path.join(tmp, "data"), "--noise", "0.01"], 300)
    assert r.returncode == 0, r.stdout[-3000:]
    r = cfg3.run(["--reference", ref, "--config", cfg_res, "--scalars", sc, "--exit-after", seconds], 900)
    assert cfg3.reached_time_limit(r), r.stdout[-3000:]
    loss, steps, times = cfg3.losses(sc)
    shutil.copy(sc, os.path.join(out, "scalars_resume.jsonl"))
    n = len(loss)
    res["resumed_from_noisy_target"] = {"iterations": n, "ms_per_iteration": cfg3.ms_per_iteration(times, steps),
                                        "loss_deciles": [sum(loss[i * n // 10:(i + 1) * n // 10]) / max(1, (i + 1) * n // 10 - i * n // 10)
                                                         for i in range(10)]}
    for i in (0, 40, 127):
        for kind in ("image", "mask"):
            shutil.copy(os.path.join(tmp, "data", kind, "%03d.png" % i), os.path.join(out, "%s_%03d.png" % (kind, i)))
    if os.environ.get("REF_NO_PROF"):
        print(json.dumps(res))
        return
    # the class-level leg once more under rocprofv3 (kernel trace + stats only)
    prof = os.path.join(out, "train_mvr_ref_prof")
    shutil.rmtree(os.path.join(tmp, "exp"), ignore_errors=True)
    sc = os.path.join(tmp, "scalars_prof.jsonl")
    if os.path.exists(sc):
        os.remove(sc)
    env_tmp = dict(os.environ, TMPDIR="/tmp")
    os.environ.update(env_tmp)
    r = cfg3.run(["--reference", ref, "--config", cfg_cls, "--scalars", sc, "--exit-after", "20"], 900,
                 prefix=["rocprofv3", "--kernel-trace", "--stats", "-d", prof, "-o", "ref", "--output-format", "csv", "--"])
    loss, steps, times = cfg3.losses(sc)
    res["profiled_iterations"] = len(loss)
    for f in glob.glob(os.path.join(prof, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, "train_mvr_ref_kernel_stats.csv"))
    for f in glob.glob(os.path.join(prof, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)   # hundreds of MB
    with open(os.path.join(out, "train_mvr_ref.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
