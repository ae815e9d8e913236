#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "backward or radius or render" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
for i in 1 2; do
  for t in 1 2 4; do
    echo "== tpw $t" >> $O/ab.txt
    BENCH_BACKWARD_TPW=$t timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
rm -rf gpurun_out/timeline
timeout 600 python tools/step_timeline.py graph > $O/timeline.txt 2>&1
timeout 600 python tools/fused_timing.py 4 > $O/fused_timing_4.txt 2>&1
timeout 1300 python -m pytest tests/test_gpu_reference_loop.py -x -q -m gpu -s -k "converges" > $O/pytest_ref5000.txt 2>&1
echo "pytest rc $?" >> $O/pytest_ref5000.txt
