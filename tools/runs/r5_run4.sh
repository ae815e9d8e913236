#!/bin/bash
# round 5, call 4: 256-segment preparation for row bands: parity + emulated per-rank step + in-kernel stamps of the BAND gather
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "band or cyclic or rank or backward" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for tpw in 0 4; do
  BAND_TPW=$tpw BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_tpw$tpw.json 2> $O/band8_tpw$tpw.err
done
timeout 300 python tools/band_fused_timing.py 8 cyclic 3 > $O/stamps_cyclic.txt 2>&1
TIMING_REBUILD=0 timeout 300 python tools/band_fused_timing.py 8 balanced 3 > $O/stamps_balanced.txt 2>&1
BAND_TRACE=1 BAND_TRACE_LAYOUT=cyclic timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o t --output-format csv -- python tools/band_timing.py 8 cfg2 > $O/trace.log 2>&1
cp $(find /tmp/prof_c -name '*kernel_stats.csv' | head -1) $O/kstats_cyclic.csv
tail -3 $O/pytest.txt; cat $O/band8_tpw*.json; cat $O/stamps_cyclic.txt $O/stamps_balanced.txt
