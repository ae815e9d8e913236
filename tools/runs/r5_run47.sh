#!/bin/bash
# round 5, call 47: the whole GPU suite + smoke + the driver's bench command on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run47; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -n 3 $O/pytest_gpu.txt; tail -n 2 $O/smoke.txt; cut -c1-200 $O/bench.json
