#!/bin/bash
# round 5, call 7: where an emulated rank's 2 ms go at configs[3] (8 x 1M points @1024^2, 8 ranks): kernel trace, both layouts
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run7; mkdir -p $O
for lay in bands cyclic; do
  BAND_TRACE=1 BAND_TRACE_LAYOUT=$lay timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$lay -o t --output-format csv -- python tools/band_timing.py 8 cfg4 > $O/trace_$lay.log 2>&1
  cp $(find /tmp/prof_$lay -name '*kernel_stats.csv' | head -1) $O/kstats_cfg4_$lay.csv
done
BAND_LAYOUTS=bands,cyclic timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4.json 2> $O/band8_cfg4.err
cat $O/band8_cfg4.json
