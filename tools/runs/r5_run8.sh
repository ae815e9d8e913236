#!/bin/bash
# round 5, call 8: large path (configs[3], 8 ranks): band filter inside the backward's cell sort -- parity + band-only kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "band or cyclic or long_list or backward" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for lay in bands cyclic; do
  BAND_TRACE=1 BAND_TRACE_LAYOUT=$lay timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$lay -o t --output-format csv -- python tools/band_timing.py 8 cfg4 > $O/trace_$lay.log 2>&1
  cp $(find /tmp/prof_$lay -name '*kernel_stats.csv' | head -1) $O/kstats_cfg4_$lay.csv
done
BAND_LAYOUTS=bands,cyclic timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4.json 2> $O/band8_cfg4.err
tail -3 $O/pytest.txt; cat $O/band8_cfg4.json
