#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run55; mkdir -p $O
timeout 185 python -m pytest tests/test_gpu_reference_loop.py -x -q -m gpu -k "converges_from_the_sphere" -s > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
