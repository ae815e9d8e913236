#!/usr/bin/env python3
"""Developer tool (BUILD CONTAINER, CPU, needs /root/reference): seed-controlled A/B of the two class stacks on the SAME CPU
oracle double, in ONE process, in lockstep.

  A = the reference's OWN classes (`DSS.core.rasterizer.SurfaceSplatting`, `EllipticalRasterizer`,
      `DSS.core.renderer.SurfaceSplattingRenderer`, unmodified) with `DSS._C` answered by the oracle double;
  B = the drop-in classes of this repository (`dss_amd.rasterizer` / `dss_amd.renderer`) with `dss_amd.ops` answered by the
      same double.

Both are built by the reference's own `config.create_model / create_trainer` from the two YAML files of
tests/ref_loop/cfg3.py, start from the same model state, and are fed the SAME batch in every iteration; the process RNG is
saved before A's `train_step` and restored before B's (the reference's `rasterizer.py:334` draws `torch.rand_like` for its
tangent basis, which shifted the batch order of round 5's one-run-each comparison).  Per iteration:

  * loss of A and of B on their own trajectories, max |points_A - points_B| (trajectory divergence);
  * PROBE (every iteration for the first `dense` iterations, every `every`-th afterwards): a third stack P = drop-in
    classes, loaded with A's state BEFORE A's step, evaluates loss + gradients at the identical state: rel-L2 of the
    position / normal gradients against A's.  If these agree to round-off along A's whole trajectory, two trajectories that
    drift apart do so by amplification of round-off (chaos), not by a systematic difference of the Python layers.

    python tools/class_level_ab.py [iterations=300] [size=96] [views=16] [points=3000] [batch=4] -> JSON lines on stdout,
    summary as the last line
"""
import copy
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_loop"))
import cfg3  # noqa: E402
import launcher  # noqa: E402

ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SIZE = int(sys.argv[2]) if len(sys.argv) > 2 else 96
VIEWS = int(sys.argv[3]) if len(sys.argv) > 3 else 16
POINTS = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
BATCH = int(sys.argv[5]) if len(sys.argv) > 5 else 4
DENSE, EVERY = 20, 10

tmp = tempfile.mkdtemp(prefix="dss_class_ab_")
cfg_cls, cfg_nat = cfg3.write_configs(tmp, size=SIZE, points=POINTS, batch=BATCH)
r = cfg3.run(["--config", cfg_cls, "--no-cuda", "--make-dataset", os.path.join(tmp, "data"), "--views", str(VIEWS),
              "--jitter", "2", "--camera-sampler"], 1800)
assert r.returncode == 0, r.stdout[-3000:]

# ---- the launcher's environment, in this process ---------------------------------------------------------------------------
for p in (os.path.join(ROOT, "compat"), ROOT, os.path.join(ROOT, "tests"), REF):
    if p not in sys.path:
        sys.path.insert(0, p)
launcher._install_stand_ins(os.devnull)
import torch  # noqa: E402
from dss_amd import ops  # noqa: E402
import oracle_ops  # noqa: E402
oracle_ops.install(ops)
os.chdir(REF)
import DSS  # noqa: E402
DSS._C = ops
sys.modules["DSS._C"] = ops
launcher._install_frnn_stand_ins()
import config  # noqa: E402  (the reference's)
import torch.optim as optim  # noqa: E402
from DSS.utils import tolerating_collate  # noqa: E402


class Stack:
    """what train_mvr.py:66-140 builds, for one YAML"""

    def __init__(self, path):
        cfg = config.load_config(path, "configs/default.yaml")
        self.cfg = cfg
        dev = torch.device("cpu")
        self.dataset = config.create_dataset(cfg.data, mode="train")
        self.model = config.create_model(cfg, camera_model=self.dataset.get_cameras(), device=dev)
        self.cameras, self.lights = self.dataset.get_cameras(), self.dataset.get_lights()
        groups = [{"params": [self.model.normals], "lr": 0.01, "betas": (0.5, 0.9)},
                  {"params": [self.model.points], "lr": 0.01, "betas": (0.5, 0.9)}]
        self.optimizer = optim.Adam(groups, lr=0.01, betas=(0.5, 0.9))
        self.scheduler = optim.lr_scheduler.MultiStepLR(self.optimizer, cfg["training"]["scheduler_milestones"],
                                                        gamma=cfg["training"]["scheduler_gamma"], last_epoch=-1)
        cfg["generation"]["resolution"] = 64
        cfg["generation"]["img_size"] = tuple(x // 4 for x in self.dataset.resolution)
        gen = config.create_generator(cfg, self.model, device=dev)
        val_loader = torch.utils.data.DataLoader(config.create_dataset(cfg.data, mode="val"), batch_size=1, shuffle=False,
                                                 collate_fn=tolerating_collate)
        self.trainer = config.create_trainer(cfg, self.model, self.optimizer, self.scheduler, gen, None, val_loader, device=dev)

    def step(self, batch, it):
        return self.trainer.train_step(copy.deepcopy(batch), cameras=self.cameras, lights=self.lights, it=it)

    def grads_only(self, batch, it):
        """`Trainer.train_step` (trainer.py:214-238) without the optimiser step"""
        t = self.trainer
        t.model.train()
        t.optimizer.zero_grad()
        if hasattr(t, "training_scheduler"):
            t.training_scheduler.step(t, it)
        data = t.process_data_dict(copy.deepcopy(batch), self.cameras, lights=self.lights)
        loss = t.compute_loss(data["img"], data["mask_img"], data["input"], data["camera"], data["light"], it=it)
        loss.backward()
        return float(loss.item())


torch.manual_seed(0)
A, B, Pb = Stack(cfg_nat), Stack(cfg_cls), Stack(cfg_cls)
B.model.load_state_dict(A.model.state_dict())
classes = {"A": "%s.%s" % (type(A.model.renderer).__module__, type(A.model.renderer).__name__),
           "B": "%s.%s" % (type(B.model.renderer).__module__, type(B.model.renderer).__name__)}
print(json.dumps({"what": "lockstep A/B of the class stacks on the CPU oracle double", "classes": classes, "iterations": ITER,
                  "image_size": SIZE, "views": VIEWS, "points": POINTS, "batch": BATCH}), flush=True)
assert classes["A"].startswith("DSS.") and classes["B"].startswith("dss_amd."), classes

gen = torch.Generator().manual_seed(0)
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
SAVE_AT = int(os.environ.get("AB_SAVE_AT", "-1"))      # save stack A's state before this iteration and stop (debugging aid)
LOAD = os.environ.get("AB_LOAD", "")                   # ... and compare the two class stacks AT that state, term by term


def scalars_of(stack, batch, it):
    """loss terms the Trainer logs + the rendered image of one forward/backward at the stack's current state"""
    seen, imgs = {}, []
    tb = stack.trainer.tb_logger
    keep = tb.add_scalar
    tb.add_scalar = lambda tag, value, step=None, *a, **k: seen.__setitem__(tag, float(value))
    ren = stack.model.renderer
    fwd = ren.forward

    def spy(*a, **k):
        out = fwd(*a, **k)
        imgs.append((out[0] if isinstance(out, tuple) else out).detach().clone())
        return out
    ren.forward = spy
    try:
        stack.grads_only(batch, it)
    finally:
        ren.forward = fwd
        tb.add_scalar = keep
    return seen, imgs


if LOAD:
    st = torch.load(LOAD, weights_only=False)   # (our own file: model state, filter object, batch)
    for S_ in (A, Pb):
        S_.model.load_state_dict(st["model"])
        S_.model.points_filter = copy.deepcopy(st["points_filter"])
    torch.set_rng_state(st["rng"])
    sa, ia = scalars_of(A, st["batch"], st["it"])
    torch.set_rng_state(st["rng"])
    sb, ib = scalars_of(Pb, st["batch"], st["it"])
    out = {"it": st["it"], "terms_A": sa, "terms_B": sb, "renders": len(ia)}
    for k, (x, y) in enumerate(zip(ia, ib)):
        d = (x - y).abs()
        out["render%d" % k] = {"shape": list(x.shape), "max_abs_diff": float(d.max()), "pixels_differing_gt_1e-4": int((d.amax(-1) > 1e-4).sum()),
                               "alpha_differs": int((x[..., -1] != y[..., -1]).sum()),
                               "per_view_pixels_differing": [int(v) for v in (d.amax(-1) > 1e-4).flatten(1).sum(1)]}
    out["grad_points_rel_l2"] = rel(Pb.model.points.grad, A.model.points.grad)
    out["grad_normals_rel_l2"] = rel(Pb.model.normals.grad, A.model.normals.grad)
    hA = getattr(A.model.renderer.rasterizer, "_Vrk_h", None)
    hB = getattr(Pb.model.renderer.rasterizer, "_Vrk_h", None)
    out["h_A"] = None if hA is None else sorted(set(round(float(v), 9) for v in hA.flatten()[:: max(1, hA.numel() // 64)]))
    out["h_B"] = None if hB is None else [float(v) for v in hB.flatten()[:16]]
    print(json.dumps(out), flush=True)
    sys.exit(0)
rows, it, t0 = [], -1, time.time()
while it + 1 < ITER:
    loader = torch.utils.data.DataLoader(A.dataset, batch_size=BATCH, shuffle=True, generator=gen, drop_last=True,
                                         collate_fn=tolerating_collate)
    for batch in loader:
        it += 1
        if it >= ITER:
            break
        row = {"it": it}
        if it == SAVE_AT:
            torch.save({"model": A.model.state_dict(), "points_filter": copy.deepcopy(A.model.points_filter), "batch": batch,
                        "it": it, "rng": torch.get_rng_state()}, os.environ.get("AB_SAVE_TO", "/tmp/ab_state.pt"))
            print(json.dumps({"saved_at": it}), flush=True)
            sys.exit(0)
        probe = (it < DENSE or it % EVERY == 0) and SAVE_AT < 0
        st = torch.get_rng_state()
        if probe:
            Pb.model.load_state_dict(A.model.state_dict())
            # (the filter state a forward leaves behind -- visibility, in-mask -- belongs to the model state too)
            Pb.model.points_filter = copy.deepcopy(A.model.points_filter)
        row["loss_A"] = A.step(batch, it)
        gA = (A.model.points.grad.clone(), A.model.normals.grad.clone())
        if probe:
            torch.set_rng_state(st)
            row["probe_loss_B_at_A_state"] = Pb.grads_only(batch, it)
            row["probe_grad_points_rel_l2"] = rel(Pb.model.points.grad, gA[0])
            row["probe_grad_normals_rel_l2"] = rel(Pb.model.normals.grad, gA[1])
            row["probe_grad_points_max_abs_diff"] = float((Pb.model.points.grad - gA[0]).abs().max())
            row["grad_points_max_abs"] = float(gA[0].abs().max())
            # what the reference's culling does at this state: points outside a camera's depth range, and the variance scale
            # h each stack derived (rasterizer.py:183-217, 320-326)
            with torch.no_grad():
                cam = A.cameras
                Vm = cam.get_world_to_view_transform().get_matrix()
                z = torch.einsum("pc,nc->np", Pb.model.points[0].detach(), Vm[:, :3, 2]) + Vm[:, 3, 2][:, None]
                zn = torch.as_tensor(cam.znear).reshape(-1, 1).float()
                zf = torch.as_tensor(cam.zfar).reshape(-1, 1).float()
                row["points_culled_per_view"] = [int(v) for v in ((z < zn) | (z > zf)).sum(1)]
                hA = getattr(A.model.renderer.rasterizer, "_Vrk_h", None)
                hB = getattr(Pb.model.renderer.rasterizer, "_Vrk_h", None)
                row["h_A_min_max"] = None if hA is None else [float(hA.min()), float(hA.max())]
                row["h_B_min_max"] = None if hB is None else [float(hB.min()), float(hB.max())]
        torch.set_rng_state(st)
        row["loss_B"] = B.step(batch, it)
        torch.set_rng_state(st)
        torch.rand(1)   # (advance the stream identically whatever the stacks drew)
        row["max_point_displacement_A_vs_B"] = float((A.model.points - B.model.points).abs().max())
        row["rel_loss_diff"] = abs(row["loss_A"] - row["loss_B"]) / max(abs(row["loss_A"]), 1e-30)
        row["t"] = round(time.time() - t0, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)

pr = [r_ for r_ in rows if "probe_grad_points_rel_l2" in r_]
summary = {
    "summary": True, "iterations": len(rows), "classes": classes, "image_size": SIZE, "views": VIEWS, "points": POINTS,
    "batch": BATCH,
    "probe_count": len(pr),
    "probe_grad_points_rel_l2_max": max(r_["probe_grad_points_rel_l2"] for r_ in pr),
    "probe_grad_normals_rel_l2_max": max(r_["probe_grad_normals_rel_l2"] for r_ in pr),
    "probe_loss_rel_diff_max": max(abs(r_["probe_loss_B_at_A_state"] - r_["loss_A"]) / max(abs(r_["loss_A"]), 1e-30) for r_ in pr),
    "rel_loss_diff_max": max(r_["rel_loss_diff"] for r_ in rows),
    "rel_loss_diff_first_above_1e-2": next((r_["it"] for r_ in rows if r_["rel_loss_diff"] > 1e-2), None),
    "max_point_displacement_final": rows[-1]["max_point_displacement_A_vs_B"],
    "loss_A_first_last": [rows[0]["loss_A"], rows[-1]["loss_A"]], "loss_B_first_last": [rows[0]["loss_B"], rows[-1]["loss_B"]],
}
print(json.dumps(summary), flush=True)
