#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run30; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "backward or median or radius" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2; do
DSS_HIP_LIBRARY=$PWD/build_ab/libdss_r4c.so timeout 300 python bench.py --timed-only 2>/dev/null | grep '^{' >> $O/ab_base.json
timeout 300 python bench.py --timed-only 2>/dev/null | grep '^{' >> $O/ab_new.json
done
timeout 300 python tools/fused_timing.py > $O/fused_timing.txt 2>&1
