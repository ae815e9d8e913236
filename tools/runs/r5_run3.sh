#!/bin/bash
# round 5, call 3: kernel traces of one emulated rank of the 8-rank step (new two-launch band path vs the round-4 library)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run3; mkdir -p $O
for lib in new r4; do
  if [ $lib = r4 ]; then export DSS_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libdss_r4final.so; else unset DSS_HIP_LIBRARY; fi
  for lay in cyclic balanced; do
    BAND_TRACE=1 BAND_TRACE_LAYOUT=$lay BAND_TPW=${TPW:-0} timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_${lib}_$lay -o t --output-format csv -- python tools/band_timing.py 8 cfg2 > $O/trace_${lib}_$lay.log 2>&1
    f=$(find /tmp/prof_${lib}_$lay -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $O/kstats_${lib}_$lay.csv
  done
done
for f in $O/kstats_*.csv; do echo "== $f"; head -14 $f | cut -d, -f1-4,7 | cut -c1-160; done
