#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run46; mkdir -p $O
timeout 150 python tools/whole_step_graph.py > $O/whole_step_graph.json 2> $O/err.txt
echo "rc $?" >> $O/err.txt
