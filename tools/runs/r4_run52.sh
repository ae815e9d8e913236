#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run52; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "cyclic" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
