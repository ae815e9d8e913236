#!/usr/bin/env python3
"""Developer tool (GPU box): same-run A/B of prebuilt library variants (DSS_HIP_LIBRARY) on tools/bench_large.py configs.
    python tools/ab_libs.py cfg3,cfg4 default build_ab/libdss_x.so ...      (alternating order, two rounds)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfgs = sys.argv[1].split(",")
libs = sys.argv[2:]
for cfg in cfgs:
    for rnd in range(2):
        for lib in (libs if rnd == 0 else libs[::-1]):
            env = dict(os.environ)
            if lib != "default":
                env["DSS_HIP_LIBRARY"] = os.path.join(ROOT, lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_large.py"), cfg], env=env, capture_output=True,
                               text=True, timeout=900)
            line = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(cfg, lib, "FAILED", r.stderr[-400:])
                continue
            d = json.loads(line[-1])
            print(cfg, lib.ljust(28), "step %.4f ms  %7.1f Msplats/s  fine %.4f  gather %s  prep %s" % (
                d["ms_per_step_eager"], d["Msplats_per_s"], d["fine_kernel_ms"], d.get("backward_gather_ms"),
                d.get("backward_prep_ms")), flush=True)
