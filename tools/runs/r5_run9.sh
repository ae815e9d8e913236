#!/bin/bash
# round 5, call 9: DSS_WS_BAND_OUTPUTS (band-only per-point outputs / sort / binning records): parity, per-rank step at
# configs[3] and at the metric's configuration, kernel trace of one rank
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py -x -q -m gpu -k "band or cyclic or long_list or backward or large_inputs" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
BAND_LAYOUTS=bands,balanced,cyclic timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4.json 2> $O/band8_cfg4.err
BAND_FULL_OUTPUTS=1 BAND_LAYOUTS=balanced timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4_fullout.json 2> $O/band8_cfg4_fullout.err
BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_cfg2.json 2> $O/band8_cfg2.err
BAND_FULL_OUTPUTS=1 BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_cfg2_fullout.json 2> $O/band8_cfg2_fullout.err
BAND_TRACE=1 BAND_TRACE_LAYOUT=balanced timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o t --output-format csv -- python tools/band_timing.py 8 cfg4 > $O/trace_balanced.log 2>&1
cp $(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $O/kstats_cfg4_balanced.csv
tail -3 $O/pytest.txt; cat $O/band8_cfg4.json $O/band8_cfg4_fullout.json $O/band8_cfg2.json $O/band8_cfg2_fullout.json
