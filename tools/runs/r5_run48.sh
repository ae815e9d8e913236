#!/bin/bash
# round 5, call 48: the RCCL world-1 tests five times over (watchdog drained before every capture), then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run48; mkdir -p $O
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_rccl_world1.py -x -q -m gpu > $O/rccl_$i.txt 2>&1; echo "rc $?" >> $O/rccl_$i.txt; tail -n 2 $O/rccl_$i.txt | tr '\n' ' '; echo; done
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -n 3 $O/pytest_gpu.txt
