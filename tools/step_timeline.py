#!/usr/bin/env python3
"""Developer tool (GPU box): timeline of one bench step from a rocprofv3 kernel trace -- per kernel duration and the gap to
the previous kernel's end, averaged over the steps of the run (graph replay or eager).
    python tools/step_timeline.py [graph|eager]"""
import csv
import glob
import os
import subprocess
import sys
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "timeline")
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
import shutil
shutil.rmtree(OUT, ignore_errors=True)   # (the traces of an earlier run in the same directory would be averaged in)
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", OUT, "--", sys.executable,
                os.path.join(ROOT, "bench.py"), "--mode", mode, "--timed-only", "--steps", "100", "--warmup", "10"],
               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
rows = []
for f in glob.glob(os.path.join(OUT, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dss::", "")[:36]))
rows.sort()
# the timed region is the longest run of back-to-back steps: find step starts = setup_bin_kernel
starts = [i for i, r in enumerate(rows) if r[2].startswith("setup_bin_kernel")]
lens = collections.Counter(b - a for a, b in zip(starts[:-1], starts[1:]))
per_step = lens.most_common(1)[0][0]     # launches per step (6 since the projection backward rides in the gather)
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:]) if b - a == per_step]
steps = steps[-100:]
acc = collections.OrderedDict()
period = []
for s, nxt in zip(steps[:-1], steps[1:]):
    period.append((nxt[0][0] - s[0][0]) / 1e3)
    prev_end = None
    for st, en, nm in s:
        d = acc.setdefault(nm, [[], []])
        d[0].append((en - st) / 1e3)
        d[1].append(0.0 if prev_end is None else (st - prev_end) / 1e3)
        prev_end = en
    acc.setdefault("(gap to next step)", [[], []])[1].append((nxt[0][0] - prev_end) / 1e3)
print("mode %s: %d steps, period mean %.2f us min %.2f" % (mode, len(period), sum(period) / len(period), min(period)))
tot_d = tot_g = 0.0
for nm, (d, g) in acc.items():
    md = sum(d) / len(d) if d else 0.0
    mg = sum(g) / len(g) if g else 0.0
    tot_d += md; tot_g += mg
    print("  %-38s dur %7.2f us   gap before %6.2f us" % (nm, md, mg))
print("  sum of durations %.2f us, sum of gaps %.2f us" % (tot_d, tot_g))
