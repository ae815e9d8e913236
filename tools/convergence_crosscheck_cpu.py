#!/usr/bin/env python3
"""Developer tool (BUILD CONTAINER, CPU, needs /root/reference): the 5,000-point sphere of `configs/dss.yml` trained by the
reference's own `train_mvr.py` for the same number of iterations through

  (a) the reference's OWN rasterizer / renderer classes (`DSS.core.rasterizer.SurfaceSplatting`, `EllipticalRasterizer`,
      `DSS.core.renderer.SurfaceSplattingRenderer`, unmodified) with `DSS._C` answered by the CPU oracle (`launcher --c-level`),
  (b) the drop-in classes of this repository with `dss_amd.ops` answered by the same oracle,

and the fraction of the model's points that end farther than 0.2 from the target surface in each.  VERDICT r4 weak 1 / ADVICE
r4: the HIP run of `tests/test_gpu_reference_loop.py` ends with ~30 % of the points driven out of the view volume while its
loss falls; the test attributes that to the reference's optimisation (Adam(lr 0.01), no pruning:
`point_modeling.py:131-132` commented out) -- this tool MEASURES whether the reference's own classes do the same.

Smaller than the GPU leg so that the CPU oracle finishes (views, resolution below; same script, same losses, same optimiser):
    python tools/convergence_crosscheck_cpu.py [iterations=1200] [size=128] [views=32] -> JSON on stdout"""
import json
import os
import sys
import tempfile

import numpy as np
import torch
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_loop"))
import cfg3  # noqa: E402

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 128
VIEWS = int(sys.argv[3]) if len(sys.argv) > 3 else 32
tmp = tempfile.mkdtemp(prefix="dss_crosscheck_")
cfg_cls, cfg_nat = cfg3.write_configs(tmp, size=SIZE, points=5000, batch=8)
r = cfg3.run(["--config", cfg_cls, "--no-cuda", "--make-dataset", os.path.join(tmp, "data"), "--views", str(VIEWS),
              "--jitter", "2", "--camera-sampler"], 1800)
assert r.returncode == 0, r.stdout[-3000:]
target = np.load(os.path.join(tmp, "data", "data_dict.npz"), allow_pickle=True)["points"].astype(np.float64)
tree = cKDTree(target)


def stats(model_pt):
    pts = torch.load(model_pt, map_location="cpu")["model"]["points"].reshape(-1, 3).double().numpy()
    d_mt, _ = tree.query(pts)
    d_tm, _ = cKDTree(pts).query(target)
    return {"model_points_farther_than_0.2": float((d_mt > 0.2).mean()), "model_to_target_median": float(np.median(d_mt)),
            "model_to_target_mean_sq": float((d_mt ** 2).mean()), "target_to_model_mean_sq": float((d_tm ** 2).mean()),
            "chamfer": float((d_mt ** 2).mean() + (d_tm ** 2).mean())}


out = {"what": "5,000-point sphere, reference train_mvr.py unmodified, CPU oracle double at the C-ABI seam", "iterations_wanted": ITER,
       "image_size": SIZE, "views": VIEWS, "target_points": int(target.shape[0])}
for name, cfg, extra in (("reference_classes_c_level", cfg_nat, ["--c-level"]), ("drop_in_classes", cfg_cls, [])):
    sc = os.path.join(tmp, "scalars_%s.jsonl" % name)
    exp = "native" if extra else "dropin"
    model_pt = os.path.join(tmp, "exp", exp, "model.pt")
    loss, legs = [], 0
    while len(loss) < ITER and legs < 60:
        legs += 1
        r = cfg3.run(["--config", cfg, "--scalars", sc, "--no-cuda", "--exit-after", "120"] + extra, 2400)
        assert cfg3.reached_time_limit(r), r.stdout[-4000:]
        loss, steps, times = cfg3.losses(sc)
        print("%s: leg %d, %d iterations" % (name, legs, len(loss)), file=sys.stderr, flush=True)
    n = len(loss)
    dec = [sum(loss[i * n // 10:(i + 1) * n // 10]) / max(1, (i + 1) * n // 10 - i * n // 10) for i in range(10)]
    out[name] = {"iterations": n, "loss_deciles": [round(x, 4) for x in dec], "at_end": stats(model_pt)}
    print(json.dumps({name: out[name]}), file=sys.stderr, flush=True)
print(json.dumps(out))
