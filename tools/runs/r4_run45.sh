#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run45; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
echo "pytest rc $?" >> $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
bash tools/collect_profiles.sh r4_d > $O/collect.log 2>&1
for c in cfg4 cfg5; do
  timeout 900 python bench.py --workload $c --no-cpu-baseline 2>$O/bench_$c.err | grep '^{' > gpurun_out/r4_d/bench_$c.json
done
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-traffic 2> $O/forced.err | grep '^{' > gpurun_out/r4_d/bench_forced_dist_world1.json
