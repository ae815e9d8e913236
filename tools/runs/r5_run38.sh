#!/bin/bash
# round 5, call 38: owner mode of the band backward (dss_render_backward_owned): parity
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_run38; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "owner_mode" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -n 30 $O/pytest.txt | cut -c1-300
