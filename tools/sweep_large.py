#!/usr/bin/env python3
"""Developer tool (GPU box): tools/bench_large.py over configurations x backward TPW settings (tools/bench_large.py reads BENCH_BACKWARD_TPW and calls dss_set_option), one summary line each."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg3", "cfg4", "cfg5"]
tpws = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"]
for c in cfgs:
    for t in tpws:
        env = dict(os.environ)
        if t != "0":
            env["BENCH_BACKWARD_TPW"] = t
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_large.py"), c], env=env, capture_output=True,
                             text=True, timeout=600)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(c, t, "FAILED", out.stderr[-300:])
            continue
        r = json.loads(line[-1])
        print(c, "tpw=" + t, "step %.4f ms" % r["ms_per_step_eager"], "%.0f Msplats/s" % r["Msplats_per_s"],
              "fine %.4f (frac %.3f)" % (r["fine_kernel_ms"], r["fine_frac_of_8TBps"]),
              "gather %s (valu frac %s)" % (r.get("backward_gather_ms"), r.get("backward_valu_frac")), flush=True)
