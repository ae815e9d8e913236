#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run53; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rccl_world1.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-traffic 2> $O/forced.err | grep '^{' > $O/bench_forced_dist_world1.json
