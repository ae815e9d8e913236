#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run10; mkdir -p $O
timeout 600 python tools/setup_timing.py > $O/setup_timing.txt 2>&1
for i in 1 2 3; do
  for lib in build_ab/libdss_r4base.so dss_amd/csrc/libdss_hip.so; do
    echo "== $lib" >> $O/ab.txt
    DSS_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
timeout 300 python bench.py --timed-only --steps 200 --mode eager >> $O/eager.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_setup.py tests/test_gpu_raster.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
