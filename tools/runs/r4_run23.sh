#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run23; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
echo "pytest rc $?" >> $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
bash tools/collect_profiles.sh r4_c > $O/collect.log 2>&1
for c in cfg4 cfg5; do
  timeout 900 python bench.py --workload $c --no-cpu-baseline 2>$O/bench_$c.err | grep '^{' > gpurun_out/r4_c/bench_$c.json
done
timeout 900 python tools/band_timing.py 8 cfg4 > gpurun_out/r4_c/band_timing_cfg4.json 2>/dev/null
