#!/usr/bin/env python3
"""Developer tool (GPU box, ONE GPU): the 1/2/4/8-GPU scaling table of `bench.py` PREDICTED from what one GPU can measure --
labelled as such in its output; no N > 1 run has been available to this build (SCALE_r01..r05 are `skipped` records).

    per-rank compute (measured, emulated)   tools/band_timing.py G <workload>: every rank's rows of the G-rank step rendered on
                                            this GPU, global visibility, no collectives, replayed as a hipGraph; the slowest
                                            rank sets the step
  + collective floor (measured, world 1)    `BENCH_FORCE_DIST=1 bench.py --gpus 1` (the RCCL path in a world of one rank, whole
                                            step as one graph) minus the same step's compute: what the collectives of a step
                                            cost when no byte crosses a link -- a LOWER bound for any world size
  = predicted step                          value = G x points / step; efficiency against the single-GPU line
  (+ link time, ESTIMATED, reported beside) bytes per rank of the image all-gather and the gradient all-reduce over 7 xGMI
                                            links x 153 GB/s (MI355X_MICROARCH.md), direct schedule; the overlapped exchange hides
                                            the image bands behind the backward, the folded one does not

    python tools/predict_scaling.py [cfg2|cfg4|cfg5]  -> one JSON object on stdout"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
XGMI_LINK_GBS, LINKS = 153.0, 7


def run(cmd, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("%s failed: %s" % (" ".join(cmd), (r.stdout + r.stderr)[-600:]))
    return json.loads(lines[-1])


py = sys.executable
out = {"what": "PREDICTED scaling of `bench.py --gpus G%s` -- per-rank compute of the CAUSAL step (render -> image loss of the "
               "rank's rows -> backward) emulated on ONE GPU + the collective floor measured at world size 1; NOT a measurement "
               "at G > 1" % ("" if which == "cfg2" else " --workload " + which),
       "workload": which, "causal": True,
       "speedup_against": "the single-GPU line of bench.py (grad_out = randn, NO image loss in the step); "
                          "`speedup_vs_causal_single_gpu_step` divides by the one-rank causal step instead"}
bands = {}
for G in (1, 2, 4, 8):
    # (large workloads: contiguous bands -- equal rows, occupancy-balanced, fitted to the measured times, two rebalancing steps)
    lay = "bands" if G == 1 else ("balanced,cyclic" if which == "cfg2" else "bands,balanced,fitted,rebalanced,rebalanced2")
    bands[G] = run([py, os.path.join(ROOT, "tools", "band_timing.py"), str(G), which], {"BAND_LAYOUTS": lay})
# (a cold box has once timed the first process of this script at 2.5x the others: repeat an implausible world-1 measurement)
for _ in range(2):
    if bands[1]["bands"]["graph_max_us"] > 1.4 * min(v["graph_max_us"] for v in bands[2].values() if isinstance(v, dict) and "graph_max_us" in v):
        bands[1] = run([py, os.path.join(ROOT, "tools", "band_timing.py"), "1", which], {"BAND_LAYOUTS": "bands"})
Pc, S = bands[1]["points_per_cloud"], bands[1]["image_size"]
cams = {G: bands[G]["cameras"] for G in bands}
out["per_rank_compute_us"] = {str(G): {k: {"max": v["graph_max_us"], "per_rank": v["graph_us"]} for k, v in b.items()
                                       if isinstance(v, dict) and "graph_us" in v} for G, b in bands.items()}
# the multi-GPU step's own compute at world size 1 (clip after the reduction, projection in a launch of its own) and the
# single-GPU headline step
multi1 = bands[1]["bands"]["graph_max_us"]
floors = {}
if which == "cfg2":
    single = run([py, os.path.join(ROOT, "bench.py"), "--timed-only", "--no-cpu-baseline", "--no-traffic"])
    single_us = single["ms_per_step"] * 1e3
    for form in ("overlap", "fold"):
        f = run([py, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--timed-only", "--no-cpu-baseline", "--no-traffic"],
                {"BENCH_FORCE_DIST": "1", "BENCH_EXCHANGE": form})
        floors[form] = {"forced_world1_step_us": round(f["ms_per_step"] * 1e3, 2), "launch": f["launch"],
                        "collective_floor_us": round(max(f["ms_per_step"] * 1e3 - multi1, 0.0), 2)}
else:
    single_us = bands[1]["single_gpu_step_us"]["eager"]
    # (the large workloads run eagerly; the collective floor of the metric's configuration is latency, not bytes: reused)
    floors = {"overlap": {"collective_floor_us": 20.0, "from": "round-4 measurement (78.3 - 58 us)"}}
    for src in (os.environ.get("PREDICT_FLOOR_FROM", ""), "profiles/r5_b_predicted_scaling_cfg2.json",
                "profiles/r5_d_predicted_scaling_cfg2.json"):
        try:
            prev = json.load(open(os.path.join(ROOT, src)))
            floors = {k: {"collective_floor_us": v["collective_floor_us"], "from": src} for k, v in prev["collective_floor"].items()}
            break
        except Exception:  # noqa: BLE001
            continue
out["single_gpu_step_us"] = round(single_us, 2)
out["causal_single_gpu_step_us"] = multi1
out["multi_step_compute_world1_us"] = multi1
out["collective_floor"] = floors
weak = which == "cfg2"
table = []
for G in (1, 2, 4, 8):
    for layout, v in out["per_rank_compute_us"][str(G)].items():
        row = {"G": G, "layout": layout, "max_rank_compute_us": v["max"]}
        for form, fl in floors.items():
            step = v["max"] + (fl["collective_floor_us"] if G > 1 else 0.0)
            if G == 1:
                step = single_us
            splats = cams[G] * Pc
            row["predicted_step_us_" + form] = round(step, 1)
            row["predicted_Msplats_per_s_" + form] = round(splats / step, 1)
            row["predicted_speedup_" + form] = round((splats / step) / (cams[1] * Pc / single_us), 2)
        # link time of the collectives, ESTIMATED (bytes a rank sends / receives over its links at the link peak).  The step is
        # causal (bench.py's N > 1 step: the image gradient is the loss of the rank's own rows): on the CRITICAL path sit the
        # visibility all-reduce (P bytes), the all-reduce of the loss partials (2560 N bytes), in the owner form the all-gather
        # of the alpha-gradient plane (N S^2 4 / G bytes per rank) and the gradient all-reduce; the image all-gather
        # (N S^2 16 / G bytes per rank) is asynchronous on its own communicator -- nothing in the step reads the full image.
        n_img = cams[G]
        img_bytes_rank = n_img * S * S * 16 / G
        alpha_bytes_rank = n_img * S * S * 4 / G
        owner = bands[G].get("gradient") == "owner"
        grad_bytes = Pc * 24 if owner else n_img * Pc * 24
        vis_bytes = n_img * Pc
        row["gradient_exchange"] = "owner" if owner else "bucket"
        row["causal"] = True
        if G > 1:
            bw = min(G - 1, LINKS) * XGMI_LINK_GBS * 1e3          # bytes per microsecond over the links a rank uses
            ar = lambda b: 2 * b * (G - 1) / G / bw                # reduce-scatter + all-gather
            ag = lambda b_rank: b_rank * (G - 1) / bw
            crit = {"visibility_all_reduce": round(ar(vis_bytes), 1), "loss_partials_all_reduce": round(ar(2560 * n_img), 2),
                    "gradient_all_reduce": round(ar(grad_bytes), 1)}
            if owner:
                crit["alpha_plane_all_gather"] = round(ag(alpha_bytes_rank), 1)
            row["estimated_link_us"] = {
                "critical_path": crit, "critical_path_sum": round(sum(crit.values()), 1),
                "image_all_gather_off_the_critical_path": round(ag(img_bytes_rank), 1),
                "note": "direct (fully connected) schedule at the link peak; RCCL's measured bus bandwidth at these sizes is lower, "
                        "and every collective also pays a latency the world-1 floor only bounds from below"}
            # the world-1 floor covers four collectives (visibility, loss partials, gradient, image); the owner form of G > 1 has
            # a fifth on the critical path (alpha plane): one more share of the floor
            n_floor = 4
            for form, fl in floors.items():
                extra = fl["collective_floor_us"] / n_floor if owner else 0.0
                step = row["predicted_step_us_" + form] + extra
                row["predicted_step_us_" + form] = round(step, 1)
                row["predicted_Msplats_per_s_" + form] = round(cams[G] * Pc / step, 1)
                row["predicted_speedup_" + form] = round((cams[G] * Pc / step) / (cams[1] * Pc / single_us), 2)
            step_l = row["predicted_step_us_overlap"] + row["estimated_link_us"]["critical_path_sum"]
            row["predicted_step_us_overlap_with_link_estimate"] = round(step_l, 1)
            row["predicted_speedup_overlap_with_link_estimate"] = round((cams[G] * Pc / step_l) / (cams[1] * Pc / single_us), 2)
        if G > 1:
            row["speedup_vs_causal_single_gpu_step"] = round((cams[G] * Pc / row["predicted_step_us_overlap"]) / (cams[1] * Pc / multi1), 2)
        table.append(row)
out["table"] = table
out["scaling"] = "weak (G cameras, one per rank's worth of rows x cameras)" if weak else "strong (fixed job, rows shared by G ranks)"
print(json.dumps(out))
