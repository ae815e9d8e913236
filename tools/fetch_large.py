#!/usr/bin/env python3
"""Developer tool (GPU box): FETCH_SIZE / WRITE_SIZE (KiB, raw) per kernel of tools/bench_large.py <cfg>."""
import csv, glob, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
res = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    OUT = os.path.join(ROOT, "gpurun_out", "fetch_large", ctr)
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", OUT, "--", sys.executable,
                    os.path.join(ROOT, "tools", "bench_large.py"), cfg], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(OUT, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:96]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][ctr] = sum(v) / len(v)
print(cfg, "per launch: FETCH MB (raw KiB x2 x1024: gfx950 correction) | WRITE MB")
for k, d in res.items():
    if "dss::" in k:
        print("  %-42s fetch %9.1f MB   write %9.1f MB" % (k, d.get("FETCH_SIZE", 0) * 2 * 1024 / 1e6, d.get("WRITE_SIZE", 0) * 1024 / 1e6))
