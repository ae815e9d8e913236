#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run28; mkdir -p $O
timeout 300 python tools/fused_timing.py > $O/fused_timing.txt 2>&1
