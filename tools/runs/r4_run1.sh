#!/bin/bash
# GPU box, round 4 run 1: dispatch-id probe, parity of the in-launch pool pass, RCCL world-1 tests, A/B against the r3 library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run1; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/dispatch_id_probe.hip -o /tmp/did 2>/dev/null && /tmp/did > $O/dispatch_id.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py tests/test_gpu_named_configs.py tests/test_gpu_image_loss.py tests/test_gpu_rccl_world1.py -x -q -m gpu > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2; do
  for lib in build_ab/libdss_r3.so dss_amd/csrc/libdss_hip.so; do
    echo "== $lib" >> $O/ab.txt
    DSS_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
timeout 600 python tools/step_timeline.py graph > $O/timeline.txt 2>&1
