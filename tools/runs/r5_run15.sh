#!/bin/bash
# round 5, call 15: eight-lane projection backward of shared clouds; exchange form chosen in the launch mode of the timed region
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_setup.py tests/test_gpu_rccl_world1.py tests/test_gpu_raster.py -x -q -m gpu -k "project or rccl or forced or two_rank or launches or training" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_cfg2.json 2> $O/band8_cfg2.err
BAND_LAYOUTS=bands timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4.json 2> $O/band8_cfg4.err
cp gpurun_out/bench_forced_dist_world1_*.json $O/ 2>/dev/null
tail -3 $O/pytest.txt; cat $O/band8_cfg2.json $O/band8_cfg4.json; python -c "
import json
for k in ('overlap','auto'):
    d=json.load(open('$O/bench_forced_dist_world1_%s.json'%k)); print(k, d['value'], d['ms_per_step'], d['config']['dist']['exchange'], d['config']['launch'])"
