#!/bin/bash
# round 5, call 19: tile rectangle computed once in the band-only setup kernel; one-thread projection backward for large shared clouds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run19; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py -x -q -m gpu -k "band or cyclic or project or two_rank" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_cfg2.json 2> $O/band8_cfg2.err
BAND_LAYOUTS=bands timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4.json 2> $O/band8_cfg4.err
BAND_TRACE=1 BAND_TRACE_LAYOUT=cyclic timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o t --output-format csv -- python tools/band_timing.py 8 cfg2 > $O/trace.log 2>&1
cp $(find /tmp/prof_c -name '*kernel_stats.csv' | head -1) $O/kstats_cfg2_cyclic.csv
tail -3 $O/pytest.txt; cat $O/band8_cfg2.json $O/band8_cfg4.json; head -8 $O/kstats_cfg2_cyclic.csv | cut -c1-120
