#!/bin/bash
# round 5, call 11: per-tile phase stamps of the fine pass on a balanced row band of configs[3] vs the whole image
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run11; mkdir -p $O
timeout 600 python tools/fine_timing.py cfg4 424,520 > $O/fine_band.txt 2>&1
timeout 600 python tools/fine_timing.py cfg4 0,224 > $O/fine_band0.txt 2>&1
cat $O/fine_band.txt; echo ======; cat $O/fine_band0.txt
