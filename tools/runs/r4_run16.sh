#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run16; mkdir -p $O
for i in 1 2 3; do
  for lib in build_ab/libdss_r4base2.so dss_amd/csrc/libdss_hip.so; do
    echo "== $lib" >> $O/ab.txt
    DSS_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
timeout 600 python tools/fused_timing.py 4 > $O/fused_timing_4.txt 2>&1
