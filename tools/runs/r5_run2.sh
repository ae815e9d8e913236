#!/bin/bash
# round 5, call 2: where the band step's time goes -- kernel traces of one emulated rank (new two-launch band path vs the
# round-4 library), tasks-per-wavefront sweep of the BAND variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5_run2; mkdir -p $O
for lib in new r4; do
  if [ $lib = r4 ]; then export DSS_HIP_LIBRARY=$PWD/build_ab/libdss_r4final.so; else unset DSS_HIP_LIBRARY; fi
  for lay in cyclic balanced; do
    (cd /tmp && BAND_TRACE=1 BAND_TRACE_LAYOUT=$lay timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_${lib}_$lay -o t -- python $GRAFT_REPO_ROOT/tools/band_timing.py 8 cfg2 > /dev/null 2>&1)
    f=$(find /tmp/prof_${lib}_$lay -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $O/kstats_${lib}_$lay.csv
  done
done
unset DSS_HIP_LIBRARY
for tpw in 1 2 4; do
  BAND_TPW=$tpw BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_tpw$tpw.json 2> $O/band8_tpw$tpw.err
done
for f in $O/kstats_*.csv; do echo "== $f"; head -12 $f | cut -c1-150; done
cat $O/band8_tpw*.json
