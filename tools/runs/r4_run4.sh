#!/bin/bash
# GPU box, round 4 run 4: bucket-sorted median + two-phase gather -- parity of each launch form, A/B, in-kernel timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4_run4; mkdir -p $O
for f in 4 5; do
  DSS_TEST_BACKWARD_FUSED=$f timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_setup.py -x -q -m gpu -k "backward or radius or render or cfg2 or fused" > $O/pytest_f$f.txt 2>&1
  echo "pytest rc $?" >> $O/pytest_f$f.txt
done
for i in 1 2; do
  for f in 1 4 5; do
    echo "== fused $f" >> $O/ab.txt
    BENCH_BACKWARD_FUSED=$f timeout 300 python bench.py --timed-only --steps 200 >> $O/ab.txt 2>&1
  done
done
for f in 4 5; do
  BENCH_BACKWARD_FUSED=$f timeout 600 python tools/step_timeline.py graph > $O/timeline_f$f.txt 2>&1
done
timeout 600 python tools/fused_timing.py 4 > $O/fused_timing_4.txt 2>&1
