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
    for name, cfg, extra in (("class_level", cfg_cls, []), ("c_level", cfg_c, ["--c-level"])):
        sc = os.path.join(tmp, "scalars_%s.jsonl" % name)
        if os.path.exists(sc):
            os.remove(sc)
        shutil.rmtree(os.path.join(tmp, "exp"), ignore_errors=True)
        r = cfg3.run(["--reference", ref, "--config", cfg, "--scalars", sc, "--exit-after", seconds] + extra, 900)
        assert cfg3.reached_time_limit(r), r.stdout[-3000:]
        loss, steps, times = cfg3.losses(sc)
        ms = cfg3.ms_per_iteration(times, steps)
        res[name] = {"iterations": len(loss), "ms_per_iteration": ms, "loss_first": loss[0], "loss_last": loss[-1],
                     "Msplats_per_s": cfg3.BATCH * cfg3.POINTS / ms * 1e-3}
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
