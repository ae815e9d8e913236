#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py tests/test_gpu_named_configs.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2 3; do
  for lib in build_ab/libdss_r4base2.so dss_amd/csrc/libdss_hip.so; do
    echo "== $lib" >> $O/ab.txt
    DSS_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
for lib in build_ab/libdss_r4base2.so dss_amd/csrc/libdss_hip.so; do
  for w in cfg3 cfg4 cfg5; do
    echo "== $w $lib" >> $O/ab_large.txt
    DSS_HIP_LIBRARY=$PWD/$lib timeout 600 python tools/bench_large.py $w 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('ms_per_step_eager', 'fine_kernel_ms', 'backward_gather_ms', 'Msplats_per_s')})" >> $O/ab_large.txt
  done
done
