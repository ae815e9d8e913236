"""GPU: the multi-GPU code path executed over RCCL (`nccl` backend) in a world of ONE rank.

No 8-GPU node has been available to this build (SCALE_r01..r03 are `skipped` records), and two ranks may not share a device
under RCCL -- but a process group of one rank on the one-GPU box still executes every line the first 8-GPU run would:
`init_process_group("nccl", device_id=...)`, the second communicator, the asynchronous `all_gather_into_tensor` of the image
bands, the visibility all-gather, the bucketed gradient all-reduce, and the three compute hipGraphs replayed beside the
process group's watchdog thread (VERDICT r3 "What's missing" 1 / "Next round" 2) -- and, since the end of round 4, the whole
step (launches AND collectives) captured as ONE hipGraph."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT",
                                                            "BENCH_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_world_of_one_reproduces_the_plain_step(tmp_path):
    """Workload(multi=True) on an `nccl` process group of one rank: eager step and graph-segment step against the plain
    single-GPU step -- image bit for bit, gradients to 1e-5 (the multi path clips after the reduction and projects in a
    separate kernel)."""
    script = os.path.join(str(tmp_path), "world1.py")
    open(script, "w").write('''
import json, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)
part = bench.RowPartition(bench.S, 1, 0)
# the reference: the same causal step (render -> image loss of the rendered rows -> backward) without any collective
img1, gw1, gc1 = [t.clone() for t in bench.Workload(dev, 1, part, multi="local").step()]
rel = lambda a, b: float((a - b).norm() / b.norm())
wl = bench.Workload(dev, 1, part, multi=True)
out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "overlap": bool(wl.fx.overlap),
       "degraded": wl.fx.degraded, "second_communicator": wl.fx.image_group is not None}
img, gw, gc = wl.step()
torch.cuda.synchronize()
out["eager"] = {"image_equal": bool(torch.equal(img, img1)), "rel_world": rel(gw, gw1), "rel_colour": rel(gc, gc1)}
wl.capture_segments()
for _ in range(3):
    img, gw, gc = wl.step_segments()
torch.cuda.synchronize()
out["graph_segments"] = {"image_equal": bool(torch.equal(img, img1)), "rel_world": rel(gw, gw1), "rel_colour": rel(gc, gc1)}
out["timing_us"] = {k: round(v, 1) for k, v in wl.dist_timing(iters=10).items()}
wl.capture_whole_step()
for _ in range(3):
    img, gw, gc = wl.step_whole()
torch.cuda.synchronize()
out["graph_step"] = {"image_equal": bool(torch.equal(img, img1)), "rel_world": rel(gw, gw1), "rel_colour": rel(gc, gc1)}
# round 5: the folded exchange (two collectives per step: the flags ride in ONE blocking image all-gather)
wl.set_exchange(True)
img, gw, gc = wl.step()
torch.cuda.synchronize()
out["fold_eager"] = {"image_equal": bool(torch.equal(img, img1)), "rel_world": rel(gw, gw1), "rel_colour": rel(gc, gc1)}
wl.capture_whole_step()
for _ in range(3):
    img, gw, gc = wl.step_whole()
torch.cuda.synchronize()
out["fold_graph_step"] = {"image_equal": bool(torch.equal(img, img1)), "rel_world": rel(gw, gw1), "rel_colour": rel(gc, gc1)}
out["fold_collectives"] = sum(1 for k, _, _ in wl.stages() if k == "x")
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
''' % ROOT)
    r = subprocess.run([sys.executable, script], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["backend"] == "nccl" and out["world_size"] == 1
    assert out["overlap"] is True and out["degraded"] is None and out["second_communicator"] is True, out
    # (world of one: visibility / image [folded into one], loss sums, gradient -- no alpha-plane exchange without a band)
    assert out["fold_collectives"] == 4
    # (graph_step: launches AND the collectives in ONE hipGraph; fold_*: the two-collective form of the exchange)
    for leg in ("eager", "graph_segments", "graph_step", "fold_eager", "fold_graph_step"):
        assert out[leg]["image_equal"], (leg, out)
        assert out[leg]["rel_world"] < 1e-5 and out[leg]["rel_colour"] < 1e-5, (leg, out)
    for k in ("wait_visibility_allreduce", "wait_loss_allreduce", "wait_gradient_allreduce", "wait_image_allgather",
              "forward_compute", "loss_sums_compute", "loss_gradient_compute", "backward_compute"):
        assert out["timing_us"][k] > 0, (k, out["timing_us"])


@pytest.mark.parametrize("exchange", ["overlap", "auto"])
def test_bench_forced_dist_runs_the_rccl_path_on_one_gpu(exchange):
    """`BENCH_FORCE_DIST=1 python bench.py --gpus 1`: the driver's command line with the multi branch forced -- ONE JSON line
    whose `config.dist` records backend nccl, the second communicator in use, graph segments and the time in each
    collective.  `overlap` (the default): the three-collective exchange; `auto`: both forms of the exchange are prepared in
    the launch mode of the timed region, timed on the ranks of the run and the faster one is kept (`config.dist.exchange`)."""
    env = _env()
    env["BENCH_FORCE_DIST"] = "1"
    env["BENCH_EXCHANGE"] = exchange
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--no-traffic"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    d = rec["config"]["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["forced"] is True
    assert d["degraded"] is None and d["segment_capture"] == "ok", d
    assert d["whole_step_graph"] == "ok", d   # (RCCL 2.26 lets its collectives be captured: the timed step is ONE graph)
    assert rec["config"]["launch"].startswith("graph_step") and rec["n_gpus"] == 1 and rec["value"] > 0
    assert d["causal"] is True and d["engine"].startswith("dss_amd.sharded.RowShardedRender"), d
    if exchange == "overlap":
        assert d["overlap"] is True and d["collectives_per_step"] == 4 and d["exchange"]["form"] == "overlap", d
        for k in ("wait_visibility_allreduce", "wait_loss_allreduce", "wait_gradient_allreduce", "wait_image_allgather",
                  "compute_us"):
            assert d["timing_us"][k]["max"] > 0, (k, d["timing_us"])
    else:
        ex = d["exchange"]
        assert ex["form"] in ("overlap", "fold") and set(ex["ms_per_step"]) == {"overlap", "fold"}, d
        assert d["collectives_per_step"] == 4, d
        assert ex["ms_per_step"][ex["form"]] == min(ex["ms_per_step"].values()), d
    try:   # keep the line for profiles/ (scratch directory of the GPU box; harmless elsewhere)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "bench_forced_dist_world1_%s.json" % exchange), "w").write(lines[0] + "\n")
    except OSError:
        pass
