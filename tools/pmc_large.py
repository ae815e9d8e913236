#!/usr/bin/env python3
"""Developer tool (GPU box): SQ counters per kernel of tools/bench_large.py <cfg> (one rocprofv3 --pmc pass)."""
import csv, glob, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
OUT = os.path.join(ROOT, "gpurun_out", "pmc_large")
ctrs = "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD".split()
os.makedirs(OUT, exist_ok=True)
subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", OUT, "--", sys.executable,
                os.path.join(ROOT, "tools", "bench_large.py"), cfg], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(OUT, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:96]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(cfg, "kernel".ljust(38), " ".join(c[3:].rjust(14) for c in ctrs))
for k, d in acc.items():
    if "dss::" not in k: continue
    print(k.ljust(72), " ".join(("%.0f" % (sum(d[c]) / max(len(d[c]), 1))).rjust(14) for c in ctrs))
