#!/usr/bin/env python3
"""Developer tool (GPU box): runs tools/fetch_calibration.hip under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and
WRITE_SIZE is not exercised: the kernels do not write) and prints, per access pattern, the raw counter (KiB x 1024) against the
bytes requested and the bytes of the distinct 32 / 64 / 128-byte granules touched -> which one FETCH_SIZE follows, and the
correction factor that pattern needs.   python tools/fetch_calibration.py > profiles/rN_fetch_calibration.txt"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = "/tmp/fetch_calibration"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "fetch_calibration.hip"), "-o", exe],
               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
out = os.path.join(ROOT, "gpurun_out", "fetch_calibration")
r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", out, "--", exe],
                   cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, check=True)
known = {}
for line in r.stdout.splitlines():
    p = line.split()
    if len(p) == 9 and p[0].startswith("cal_"):
        known[p[0]] = {"requested": int(p[2]), "g32": int(p[4]), "g64": int(p[6]), "g128": int(p[8])}
counter = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE":
            name = row["Kernel_Name"].split("(")[0]
            counter[name] = counter.get(name, 0.0) + float(row["Counter_Value"]) * 1024.0
print("pattern          FETCH_SIZE raw B    requested B   distinct 32-B    64-B        128-B     | raw/requested  raw/g32  raw/g64  raw/g128")
for k, v in known.items():
    c = counter.get(k)
    if c is None:
        print(k, "no counter sample")
        continue
    print("%-16s %14.0f %14d %14d %11d %11d | %8.3f %10.3f %8.3f %8.3f" % (
        k, c, v["requested"], v["g32"], v["g64"], v["g128"], c / v["requested"], c / v["g32"], c / v["g64"], c / v["g128"]))
print("(a pattern whose raw counter is ~0.5 of a granule column needs the x2 of MI355X_MICROARCH.md for THAT granule size; ~1.0 needs none)")
