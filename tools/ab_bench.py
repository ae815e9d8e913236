#!/usr/bin/env python3
"""Developer tool (GPU box): same-run A/B of prebuilt library variants (DSS_HIP_LIBRARY) on the DRIVER's command,
`bench.py --timed-only` (the metric's configuration unless --workload is given), alternating order, three rounds.
    python tools/ab_bench.py [--workload cfg4] [--env K=V ...] default build_ab/libdss_x.so ...
A variant may carry its own environment:  build_ab/libdss_x.so@BENCH_ORDER_REFRESH=16"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra, common = [], {}
while args and args[0].startswith("--"):
    if args[0] == "--workload":
        extra += ["--workload", args[1]]
    elif args[0] == "--env":
        k, v = args[1].split("=", 1)
        common[k] = v
    args = args[2:]
res = {}
for rnd in range(3):
    for spec in (args if rnd % 2 == 0 else args[::-1]):
        lib, _, envs = spec.partition("@")
        env = dict(os.environ)
        env.update(common)
        for kv in filter(None, envs.split(",")):
            k, v = kv.split("=", 1)
            env[k] = v
        if lib != "default":
            env["DSS_HIP_LIBRARY"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--timed-only", "--no-cpu-baseline", "--no-traffic"] + extra,
                           env=env, capture_output=True, text=True, timeout=1200)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(spec, "FAILED", r.stderr[-400:], flush=True)
            continue
        d = json.loads(line[-1])
        res.setdefault(spec, []).append(d["ms_per_step"])
        print("round %d  %-60s %.5f ms/step  %8.1f Msplats/s  (%s)" % (rnd, spec, d["ms_per_step"], d["value"], d["launch"]), flush=True)
for spec, v in res.items():
    print("%-60s min %.5f  median %.5f ms/step" % (spec, min(v), sorted(v)[len(v) // 2]))
