#!/bin/bash
# round 5, call 13: checkpoint -- the whole GPU suite, smoke, the driver's bench line, configs[3]/[4] lines, band steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run13; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
for c in cfg4 cfg5; do timeout 600 python bench.py --workload $c --no-cpu-baseline --no-traffic > $O/bench_$c.json 2> $O/bench_$c.err; done
BAND_LAYOUTS=bands,balanced,cyclic timeout 600 python tools/band_timing.py 8 cfg4 > $O/band8_cfg4.json 2> $O/band8_cfg4.err
BAND_LAYOUTS=balanced,cyclic timeout 300 python tools/band_timing.py 8 cfg2 > $O/band8_cfg2.json 2> $O/band8_cfg2.err
tail -3 $O/pytest.txt; tail -2 $O/smoke.txt; cat $O/band8_cfg4.json $O/band8_cfg2.json; for c in "" _cfg4 _cfg5; do python -c "
import json,sys
d=json.loads([l for l in open('$O/bench$c.json') if l.startswith('{')][-1])
print('$c', d['value'], d['ms_per_step'], d.get('value_with_knn'), d.get('value_via_api'))"; done
