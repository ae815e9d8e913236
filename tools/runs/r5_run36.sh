#!/bin/bash
# round 5, call 36: the whole GPU suite, smoke, and the round's final profile set (tools/collect_profiles.sh r5_c)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run36; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 3000 bash tools/collect_profiles.sh r5_c > $O/collect.log 2>&1
tail -n 3 $O/pytest_gpu.txt; tail -n 2 $O/smoke.txt; cut -c1-300 gpurun_out/r5_c/bench.json
