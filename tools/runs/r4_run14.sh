#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run14; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "backward or radius or render or cfg2" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2 3; do
  for lib in build_ab/libdss_r4base2.so dss_amd/csrc/libdss_hip.so; do
    for t in 2 1; do
      echo "== $lib tpw=$t" >> $O/ab.txt
      BENCH_BACKWARD_TPW=$t DSS_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
    done
  done
done
timeout 600 python tools/fused_timing.py 4 > $O/fused_timing_4.txt 2>&1
