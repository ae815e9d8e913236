#!/bin/bash
# round 5, call 1: band path on the two-launch backward (BAND variants) -- parity tests + same-box A/B of the emulated
# per-rank step against the round-4 library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5_run1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "band or cyclic or rank or backward" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for lib in new r4; do
  if [ $lib = r4 ]; then export DSS_HIP_LIBRARY=$PWD/build_ab/libdss_r4final.so; else unset DSS_HIP_LIBRARY; fi
  timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_$lib.json 2> $O/band8_$lib.err
done
unset DSS_HIP_LIBRARY
for G in 2 4; do timeout 300 python tools/band_timing.py $G cfg2 > $O/band${G}_new.json 2> $O/band${G}_new.err; done
timeout 200 python bench.py --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.txt; cat $O/band8_new.json $O/band8_r4.json $O/band2_new.json $O/band4_new.json; tail -c 600 $O/bench.json
