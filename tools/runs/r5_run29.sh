#!/bin/bash
# round 5, call 29: window sizes of the backward at the bench workloads (why the one-column-slot path of the gather did not pay)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run29; mkdir -p $O
for w in headline cfg3 cfg4 cfg5; do timeout 300 python tools/window_stats.py $w 2>/dev/null | tee -a $O/window_stats.jsonl; done
