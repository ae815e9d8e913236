#!/bin/bash
# round 5, call 51: owner mode at image sizes that are not powers of two and with five feature channels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run51; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "owner_mode" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -n 25 $O/pytest.txt | cut -c1-250
