#!/bin/bash
# round 5, call 41: the whole GPU suite, smoke, and the final profile set after the last kernel changes (tools/collect_profiles.sh r5_d)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run41; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 3000 bash tools/collect_profiles.sh r5_d > $O/collect.log 2>&1
tail -n 3 $O/pytest_gpu.txt; tail -n 2 $O/smoke.txt; cut -c1-200 gpurun_out/r5_d/bench.json
