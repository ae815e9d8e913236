import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "ref_loop"))
import cfg3
tmp = "/tmp/train_mvr_ref"; os.makedirs(tmp, exist_ok=True)
ref = cfg3.reference_root(tmp)
cfg_cls, cfg_c = cfg3.write_configs(tmp)
if not os.path.isdir(os.path.join(tmp, "data", "image")):
    r = cfg3.run(["--reference", ref, "--config", cfg_cls, "--make-dataset", os.path.join(tmp, "data"), "--views", str(cfg3.VIEWS), "--jitter", str(cfg3.JITTER), "--camera-sampler"], 900)
    assert r.returncode == 0, r.stdout[-2000:]
os.environ["DSS_REF_LOOP_CPROFILE"] = os.path.join(ROOT, "gpurun_out", "ref_loop_host_profile.txt")
r = cfg3.run(["--reference", ref, "--config", cfg_cls, "--scalars", os.path.join(tmp, "sc.jsonl"), "--exit-after", "25"], 900)
print(r.stdout[-300:])
