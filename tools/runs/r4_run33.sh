#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run33; mkdir -p $O
timeout 600 python tools/api_threads.py > $O/api_threads.json 2> $O/err.txt
