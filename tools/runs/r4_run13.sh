#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run13; mkdir -p $O
timeout 600 python tools/fused_timing.py 4 > $O/fused_timing_4.txt 2>&1
