#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run22; mkdir -p $O
cd /tmp
BAND_TRACE=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/ks -o b --output-format csv -- python $GRAFT_REPO_ROOT/tools/band_timing.py 8 cfg2 > $GRAFT_REPO_ROOT/$O/band_trace.json 2>$GRAFT_REPO_ROOT/$O/band_trace.err
cp $(find $GRAFT_REPO_ROOT/$O/ks -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/$O/kernel_stats_band_cyclic_rank3.csv; rm -rf $GRAFT_REPO_ROOT/$O/ks
cd $GRAFT_REPO_ROOT
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-traffic 2> $O/forced.err | grep '^{' > $O/bench_forced_dist_world1.json
BENCH_FORCE_DIST=1 BENCH_IMAGE_LATE=1 timeout 600 python bench.py --no-cpu-baseline --no-traffic 2>> $O/forced.err | grep '^{' > $O/bench_forced_dist_world1_late.json
