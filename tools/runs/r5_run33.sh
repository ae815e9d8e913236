#!/bin/bash
# round 5, call 33: + the visible points of a band rank start from zero in visible_scan (no scattered zero stores in cell_hist): parity, rank split
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run33; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_named_configs.py tests/test_gpu_point_order.py tests/test_gpu_raster.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for c in cfg4 cfg5; do
  BAND_LAYOUTS=bands,balanced timeout 900 python tools/band_timing.py 8 $c > $O/band8_$c.json 2> $O/band8_$c.err
  BAND_TRACE=1 BAND_TRACE_LAYOUT=bands BAND_TRACE_RANK=3 rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -o b --output-format csv -- python tools/band_timing.py 8 $c > /dev/null 2>&1
  cp $(find /tmp/ks_$c -name '*kernel_stats.csv' | head -1) $O/band_kernel_stats_${c}_bands_rank3.csv
done
tail -n 3 $O/pytest.txt; cat $O/band8_*.json | cut -c1-700
