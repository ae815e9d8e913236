#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run34; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
