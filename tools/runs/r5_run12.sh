#!/bin/bash
# round 5, call 12: 16-byte scan of the pool pass's mask bytes; transposed position store of the band-only setup
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_named_configs.py tests/test_gpu_point_order.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
BAND_LAYOUTS=bands,balanced,cyclic timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4.json 2> $O/band8_cfg4.err
for c in cfg4 cfg5; do timeout 600 python bench.py --workload $c --no-cpu-baseline --no-traffic > $O/bench_$c.json 2> $O/bench_$c.err; done
BAND_TRACE=1 BAND_TRACE_LAYOUT=balanced timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o t --output-format csv -- python tools/band_timing.py 8 cfg4 > $O/trace_balanced.log 2>&1
cp $(find /tmp/prof_b -name '*kernel_stats.csv' | head -1) $O/kstats_cfg4_balanced.csv
tail -3 $O/pytest.txt; cat $O/band8_cfg4.json; for c in cfg4 cfg5; do python -c "
import json,sys
d=json.loads([l for l in open('$O/bench_$c.json') if l.startswith('{')][-1])
print('$c', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel_ms'), d.get('roofline_other',{}).get('kernel_ms'))"; done
