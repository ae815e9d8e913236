#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run43; mkdir -p $O
timeout 900 python tools/knn_sweep.py > $O/knn_sweep.json 2> $O/err.txt
