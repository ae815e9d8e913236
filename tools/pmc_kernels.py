#!/usr/bin/env python3
"""Developer tool (GPU box): SQ issue/wait counters per kernel of the bench step (one rocprofv3 --pmc pass).
WAIT_ANY (parked on s_waitcnt/barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES
(MI355X_MICROARCH.md, SQ counters; units are quad-cycles)."""
import csv, glob, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc_sq")
ctrs = "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD".split()
os.makedirs(OUT, exist_ok=True)
cmd = ["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", OUT, "--",
       sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "eager", "--no-cpu-baseline", "--no-traffic", "--steps", "20", "--warmup", "5"]
subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(OUT, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:96]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel".ljust(72), " ".join(c[3:].rjust(14) for c in ctrs))
for k, d in acc.items():
    if "dss::" not in k: continue
    print(k.ljust(72), " ".join(("%.0f" % (sum(d[c]) / max(len(d[c]), 1))).rjust(14) for c in ctrs))
