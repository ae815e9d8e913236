#!/bin/bash
# round 5, call 16: phase stamps inside setup_bin_kernel at the metric's configuration (inputs | arithmetic | stores | atomics)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run16; mkdir -p $O
timeout 600 python tools/setup_timing.py > $O/setup_timing.txt 2>&1
tail -12 $O/setup_timing.txt
