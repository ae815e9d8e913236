#!/usr/bin/env python3
"""Developer tool (GPU box): HBM traffic of the dominant kernel from rocprofv3 PMC counters.

Runs bench.py under rocprofv3 twice -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, with
--kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE costs 3 TCC slots,
WRITE_SIZE 2; they do not fit one pass) -- and averages the counters over the fine_kernel
dispatches.  Corrections per MI355X_MICROARCH.md "HBM": both counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is taken as is
(uncalibrated).  Prints one JSON line (bench.py spawns this script and reports the result as roofline.traffic, bytes per
launch) and leaves a copy under gpurun_out/ for tools/collect_profiles.sh."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# `--workload cfg4|cfg5`: the same passes over `bench.py --workload ...` (eager launches, the renderer's cached point order
# inside the timed region): the line of the large workloads then carries its own counters and rocprofv3 averages
WORKLOAD = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "cfg2"
LARGE = WORKLOAD != "cfg2"
WL_ARGS = ["--workload", WORKLOAD] if LARGE else []
OUT = os.path.join(ROOT, "gpurun_out", "traffic" + ("_" + WORKLOAD if LARGE else ""))
KERNEL = "fine_kernel"
OTHER = "render_backward_kernel"   # the backward gather: reported next to it (same passes)
OTHER_MEAN = {}


def one_pass(counter):
    d = os.path.join(OUT, counter)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "eager", "--timed-only", "--steps", "16" if LARGE else "20",
           "--warmup", "2" if LARGE else "5"] + WL_ARGS
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    vals, other = [], []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            if KERNEL in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
            elif OTHER in r["Kernel_Name"]:
                other.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s samples for %s" % (counter, KERNEL))
    OTHER_MEAN[counter] = sum(other) / len(other) if other else None
    return sum(vals) / len(vals), len(vals)


def trace_pass():
    """Third pass, kernel trace only (no counters: a PMC pass serialises and perturbs the kernels): the rocprofv3 average
    duration of both roofline kernels inside the very step bench.py times -> {kernel: ms}"""
    d = os.path.join(OUT, "trace")
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "eager" if LARGE else "graph", "--timed-only", "--steps",
           "16" if LARGE else "100", "--warmup", "2" if LARGE else "10"] + WL_ARGS
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            for k in (KERNEL, OTHER):
                if k in r["Kernel_Name"]:
                    acc.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return {k: sum(v) / len(v) for k, v in acc.items() if v}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, nf = one_pass("FETCH_SIZE")
    write, nw = one_pass("WRITE_SIZE")
    try:
        prof_ms, prof_n = trace_pass()
    except Exception:  # noqa: BLE001  (the traffic numbers stand on their own)
        prof_ms, prof_n = {}, {}
    rec = {"kernel": "fine_kernel<5>", "workload": WORKLOAD,
           "command": "bench.py --mode eager --timed-only --steps %s%s" % ("16" if LARGE else "20", " --workload " + WORKLOAD if LARGE else " (BASELINE configs[1])"),
           "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write, "samples": [nf, nw],
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as reported",
           "traffic_bytes_per_launch": int((2.0 * fetch + write) * 1024)}
    if OTHER_MEAN.get("FETCH_SIZE") is not None and OTHER_MEAN.get("WRITE_SIZE") is not None:
        rec["render_backward_kernel_traffic_bytes_per_launch"] = int((2.0 * OTHER_MEAN["FETCH_SIZE"] + OTHER_MEAN["WRITE_SIZE"]) * 1024)
    rec["rocprof_kernel_ms"] = prof_ms
    rec["rocprof_kernel_dispatches"] = prof_n
    rec["rocprof_kernel_ms_how"] = "rocprofv3 --kernel-trace (no counters) over bench.py --mode graph --steps 100: mean End - Start"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "traffic_fine_kernel%s.json" % ("_" + WORKLOAD if LARGE else "")), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
