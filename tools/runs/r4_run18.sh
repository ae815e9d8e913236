#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run18; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
echo "pytest rc $?" >> $O/pytest_all.txt
cp gpurun_out/bench_forced_dist_world1.json $O/ 2>/dev/null
timeout 900 python tools/band_timing.py 8 cfg2 > $O/band_timing_cfg2.json 2> $O/band_timing_cfg2.err
