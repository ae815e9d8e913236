#!/bin/bash
# round 5, call 5: BAND gather with packed cloud ids / rs in LDS / median priority; batched project_backward with the colour
# reduction; row-bias sweep of the balanced bounds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py -x -q -m gpu -k "band or cyclic or rank or backward" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for tpw in 0 4; do
  BAND_TPW=$tpw BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_tpw$tpw.json 2> $O/band8_tpw$tpw.err
done
for bias in 100 300; do
  BAND_ROW_BIAS=$bias BAND_LAYOUTS=balanced timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_bias$bias.json 2> $O/band8_bias$bias.err
done
timeout 300 python tools/band_fused_timing.py 8 balanced 3 > $O/stamps_balanced.txt 2>&1
TIMING_REBUILD=0 BAND_TPW=4 timeout 300 python tools/band_fused_timing.py 8 cyclic 3 > $O/stamps_cyclic_tpw4.txt 2>&1
BAND_TRACE=1 BAND_TRACE_LAYOUT=balanced timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o t --output-format csv -- python tools/band_timing.py 8 cfg2 > $O/trace.log 2>&1
cp $(find /tmp/prof_c -name '*kernel_stats.csv' | head -1) $O/kstats_balanced.csv
tail -3 $O/pytest.txt; cat $O/band8_*.json; tail -22 $O/stamps_balanced.txt;  tail -12 $O/stamps_cyclic_tpw4.txt
